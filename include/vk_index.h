/*
 * vk_index.h -- C ABI of the MI355X-native vector-kNN backend (libvkindex.so).
 *
 * Drop-in boundary for valkey-search's vector hot path.  The reference has no
 * FFI on this path: `VectorFlat<float>` / `VectorHNSW<float>`
 * (src/indexes/vector_flat.cc, vector_hnsw.cc) own an `hnswlib::BruteforceSearch`
 * / `hnswlib::HierarchicalNSW` object and call its C++ members.  This header
 * replaces exactly that seam -- the hnswlib `AlgorithmInterface`
 * (third_party/hnswlib/hnswlib.h:215-235) plus the members valkey-search reaches
 * into -- with a plain C ABI (pointers and sizes only), so the host classes that
 * sit above it (valkey-search_amd/csrc/host/, or the reference's own classes via
 * the adaptor shown in INTEGRATION.md) stay ordinary C++.  Each entry point cites
 * the reference interface it stands in for (paths relative to the reference tree).
 *
 * Conventions
 *  - every function returns a vk_status; the message of the last failure on the
 *    calling thread is vk_last_error().  No C++ exception crosses the ABI (the
 *    reference turns every hnswlib exception into absl::InternalError at this
 *    seam: vector_flat.cc:68-72,165-176,238-242; vector_hnsw.cc:102-106,186-197,331-335).
 *  - rows/queries are `dim` elements of the index dtype (f32 = 4*dim bytes), borrowed
 *    for the call only; the library copies what it keeps (the reference keeps the
 *    caller's pointer: bruteforce.h:81, hnswalg.h:1576-1577).
 *  - COSINE: as in the reference (vector_base.cc:61-76,140-150) the caller
 *    normalises rows and queries; the library then works in the inner-product space.
 *  - labels are the u64 "internal ids" VectorBase hands to hnswlib (vector_base.cc:340-358): a counter from zero.
 *    UINT64_MAX is reserved -- it is the padding label of result lists -- and refused with VK_ERR_INVALID by every call
 *    that brings a label in (add, add_batch, the device bulk load, a stream being loaded).
 *  - results are ascending by (distance, label) -- the order VectorBase::CreateReply
 *    produces from hnswlib's heap (vector_base.cc:258-277).
 *  - all entry points are thread safe; searches may run concurrently, mutations are
 *    staged on the host and published to HBM by vk_index_flush() (implicitly before
 *    the next search), matching the reference's reader/writer phases
 *    (vmsdk/src/time_sliced_mrmw_mutex.h:42-52).
 *  - the library needs a gfx950 device; without one vk_index_create fails with
 *    VK_ERR_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef VK_INDEX_H_
#define VK_INDEX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vk_index vk_index;

typedef enum {
  VK_OK = 0,
  VK_ERR_INVALID = 1,    /* bad argument (absl::InvalidArgumentError) */
  VK_ERR_CAPACITY = 2,   /* "The number of elements exceeds the specified limit" (bruteforce.h:74, hnswalg.h:1551) */
  VK_ERR_NOT_FOUND = 3,  /* unknown / tombstoned label */
  VK_ERR_INTERNAL = 4,   /* absl::InternalError: device or library failure */
  VK_ERR_CANCELLED = 5,  /* search cancelled and partial results not wanted (vector_hnsw.cc:327-329) */
  VK_ERR_NO_DEVICE = 6,  /* no usable gfx950 device */
  VK_ERR_BUSY = 7        /* vk_index_search_submit: the query queue is full (max-query-queue-depth,
                            src/valkey_search_options.cc:231-234: FT.SEARCH is rejected before it is queued) */
} vk_status;
#define VK_STATUS_COUNT 8

typedef enum { VK_ALGO_FLAT = 0, VK_ALGO_HNSW = 1 } vk_algo;          /* IndexerType kFlat / kHNSW */
typedef enum { VK_METRIC_L2 = 0, VK_METRIC_IP = 1, VK_METRIC_COSINE = 2 } vk_metric; /* index_schema.proto DistanceMetric */
/* Storage type of the rows in HBM.  FLOAT32 is the only reference type (vector_base.h:112-114).
 * BF16 is an extension (BASELINE config 4): rows still ARRIVE as f32 and are rounded to nearest-even
 * at ingest; queries stay f32; distances are the same f32 arithmetic on the widened values, so a bf16
 * index answers exactly like an f32 index holding the rounded rows.  get_row / save return f32. */
typedef enum { VK_DTYPE_F32 = 0, VK_DTYPE_BF16 = 1 } vk_dtype;

#define VK_MAX_SHARDS 16

typedef struct vk_index_params {
  uint32_t struct_size;       /* = sizeof(vk_index_params) */
  uint32_t algo;              /* vk_algo */
  uint32_t metric;            /* vk_metric */
  uint32_t dtype;             /* vk_dtype (storage type on the device) */
  uint32_t dim;               /* VectorIndex.dimension_count */
  uint32_t block_size;        /* FLAT BLOCK_SIZE / hnsw-block-size: informational, growth is vk_index_resize */
  uint64_t initial_cap;       /* VectorIndex.initial_cap == max_elements of the hnswlib ctor */
  uint32_t m;                 /* HNSW M (hnswalg.h:121) */
  uint32_t ef_construction;   /* HNSW efC */
  uint32_t ef_runtime;        /* HNSW default ef (setEf, vector_hnsw.cc:98) */
  uint32_t allow_replace_deleted; /* hnsw-allow-replace-deleted (vector_hnsw.cc:99-100) */
  uint64_t random_seed;       /* hnswlib ctor random_seed, reference default 100 */
  int32_t device_id;          /* HIP device ordinal, -1 = current device */
  uint32_t build_threads;     /* HNSW: host threads used by vk_index_add_batch, 0 = hardware */
  /* Multi-GPU (one node, one process): n_shards >= 1 makes ONE index of n_shards sub-indexes, shard s on HIP device
   * shard_devices[s] (-1 = current device; a device may be named more than once: "logical shards").  Rows are dealt
   * to the shards in contiguous runs, every search fans out to all shards and the per-shard top-k lists are merged by
   * (distance,label) on the first shard's device -- the role of the cluster fan-out and
   * SearchPartitionResultsTracker::AddResult (src/query/fanout.cc:162-175) inside one index; HNSW = one graph per
   * shard, like one per cluster shard.  0 = a plain single-device index on device_id. */
  uint32_t n_shards;
  int32_t shard_devices[VK_MAX_SHARDS];
  /* Sharded HNSW: every shard holds an independent graph over 1/n_shards of the rows and is searched with
   * ef * shard_ef_pct / 100 (at least k), 0 = 100.  Searching every shard with the full ef costs n_shards times the hops of
   * one graph for a merged recall ABOVE the single graph's ("may scale sub-linearly", docs/topics/search.md:84); the
   * percentage that brings the merged recall back to the single graph's is a property of the data (bench.py measures
   * it: "matched-recall" against "matched-ef"). */
  uint32_t shard_ef_pct;
  /* vk_index_load only: 1 = the reference's kill switch `hnsw-validation-enable no` (src/valkey_search_options.cc:
   * 156-162, hnswalg.h:872-885 loadCheck) for a validation rule that rejects a stream the field needs to load.  The
   * reference then skips EVERY check; this library still refuses what its device kernels could not survive (sizes,
   * counts, neighbour ids out of range, links to levels a node does not have) and skips the graph-invariant checks
   * that are not memory-unsafe: self-loops, mult against M, the entry point's level. */
  uint32_t load_skip_validation;
} vk_index_params;

/* vk_index_get_stats fills sizeof(vk_index_stats) bytes of THIS header's layout: the struct grows at its end from round to round
 * (vk_index_params carries struct_size and is checked; this one is output only), so a binding is rebuilt against the header of
 * the library it loads -- tests/test_abi_symbols.py holds the Python binding and the in-tree binaries to that. */
typedef struct vk_index_stats {
  uint64_t count;             /* cur_element_count_ (live + tombstoned) */
  uint64_t deleted;           /* num_deleted_ (HNSW tombstones) */
  uint64_t capacity;          /* GetCapacity(): data_->getCapacity() / max_elements_ */
  uint64_t device_bytes;      /* HBM held by the index */
  uint64_t host_bytes;        /* host memory held by the library */
  uint64_t staged_ops;        /* mutations not yet published to the device */
  int32_t max_level;          /* HNSW maxlevel_ (-1 when empty) */
  uint32_t entry_point;       /* HNSW enterpoint_node_ */
  /* work counters of the most recent search batch (HNSW layer 0; hnswlib only counts
   * upper layers, hnswalg.h:1679-1680): */
  uint64_t last_n_eval;       /* distance evaluations */
  uint64_t last_n_hops;       /* expanded nodes */
  /* filtered / tombstoned HNSW searches keep the whole frontier (hnswlib's unbounded candidate_set, hnswalg.h:367-370)
   * in HBM: queries of the most recent batch whose frontier outgrew the first launch's 64k entries and were answered
   * by the launch with the graph-sized frontier; and entries dropped (always 0: a search is never truncated -- a
   * non-zero count fails the call with VK_ERR_INTERNAL) */
  uint64_t last_frontier_redo;
  uint64_t last_frontier_dropped;
  /* batched FLAT searches through the f16 candidate filter + exact re-rank (host entry points only): survivors of the
   * most recent batch summed over its queries (0 = the batch did not take that path), and whether a survivor list
   * overflowed so that the exact matrix-core kernel answered instead */
  uint64_t last_filter_candidates;
  uint64_t last_filter_fallback;
  /* cumulative: batches that went through the candidate filter, and -- while the option kernel-timing is 1 (off by
   * default: two event records per batch) -- the device time of their final-pass launches (HIP events on the stream they
   * ran on): with the option on over a window, delta(filter_kernel_ns) / delta(filter_batches) is the per-launch
   * duration behind bench.py's HBM roofline figure.  A sharded index reports min(batches) / max(ns) over its shards;
   * per-shard deltas come from vk_index_shard_stats */
  uint64_t filter_batches;
  uint64_t filter_kernel_ns;
  /* query coalescer (vk_index_set_coalescing): device batches run / single queries they carried */
  uint64_t coalesced_batches;
  uint64_t coalesced_queries;
  /* sharded index (n_shards >= 1), cumulative: searches fanned out to the shards and the HOST time their fan-out took
   * (enqueue of every shard's work + gather + merge launch, one enqueue thread per DEVICE; no device time) */
  uint64_t fanout_calls;
  uint64_t fanout_enqueue_ns;
  /* ---- cumulative since creation: what the adaptor's RespondWithInfoImpl (src/indexes/vector_base.cc:385-409) and the
   * module's metrics (src/metrics.h:40-50 query counters, :75-80 hnsw/flat exception counters; latency samples taken at
   * src/query/search.cc:149,160) are fed from ---------------------------------------------------------------------- */
  uint64_t searches;                       /* queries answered, all entry points */
  uint64_t search_calls;                   /* ABI search calls / dispatcher batches that carried them */
  uint64_t search_errors[VK_STATUS_COUNT]; /* failed search calls by vk_status ([VK_ERR_CANCELLED] = timeouts, ...) */
  uint64_t total_n_eval;                   /* HNSW: distance evaluations (metric_distance_computations, hnswalg.h:98) */
  uint64_t total_n_hops;                   /* HNSW: expanded nodes (metric_hops, hnswalg.h:99) */
  uint64_t tombstoned_bytes;               /* HNSW: rows + link lists of mark-deleted elements (reclaimable_memory, hnswalg.h:1199) */
  /* latency of a search as its caller saw it (call -> answer in host memory; through the dispatcher: submit -> completion),
   * per QUERY: bucket 0 = under 64 us, bucket i = [64 << (i-1), 64 << i) us, the last bucket is open-ended (>= 1 s) */
  uint64_t latency_hist[16];
  uint64_t latency_sum_ns;
  /* dispatcher (vk_index_search_submit / coalesced vk_index_search): requests accepted, rejected with VK_ERR_BUSY, waiting
   * right now, and the largest number of device batches that were in flight at once */
  uint64_t submitted;
  uint64_t rejected;
  uint64_t queued_now;
  uint64_t max_batches_in_flight;
  /* sharded index: fan-outs whose per-shard lists were gathered by the RCCL all-gather (option shard-gather = 1) instead of
   * peer copies */
  uint64_t rccl_gathers;
  /* batched FLAT through the candidate filter, most recent batch (host entry points): survivors that passed the re-rank's
   * second bound -- the rows that actually got an exact distance -- summed over the queries */
  uint64_t last_filter_reranked;
  /* the largest label the index has ever held (0 when empty): VectorBase resumes its id counter from it after an RDB load
   * (GetMaxInternalLabel, vector_hnsw.cc:387-394; vector_base.cc:480-481) */
  uint64_t max_label;
  /* dispatcher: members of device batches that were answered (or left) BEFORE their batch finished because their own token
   * went up; device-resident filters: built, served from the cache, looked up and not found, resident right now (count, bytes) */
  uint64_t cancelled_early;
  uint64_t filters_built;
  uint64_t filter_cache_hits;
  uint64_t filter_cache_misses;
  uint64_t filter_cache_entries;
  uint64_t filter_cache_bytes;
  /* HNSW: vk_index_add calls that were staged and linked later in bulk (at vk_index_flush / the next search), and how many of
   * those went through the device build (K9) rather than the host builder */
  uint64_t staged_adds;
  uint64_t staged_adds_device;
  /* HNSW: how the most recent search launch kept its visited set: 0 = one bit per node in memory, 1 = hash table in memory,
   * 2 = buckets in memory with counts in LDS, 3 = the 12 KB set in LDS (spill to memory), 5 = the 32 KB set in LDS
   * (option hnsw-visited-mode picks among what fits; bench.py names the kernel it timed from this, not from ef) */
  uint64_t last_visited_mode;
  /* the dispatcher's runner threads, summed over the runners, in microseconds since the index was created: waiting for
   * requests, inside the batching window, inside the batch's search (upload + kernels + download: two batches in flight
   * overlap on the device, so this can exceed wall time), handing answers out themselves; and the completer threads' time
   * inside the callers' completion callbacks (option completer-threads) */
  uint64_t dispatch_idle_us;
  uint64_t dispatch_window_us;
  uint64_t dispatch_search_us;
  uint64_t dispatch_handout_us;
  uint64_t dispatch_completer_us;
  /* FLAT, batched path: rows the MAIN pass of the most recent batch walked (the launch option kernel-timing brackets): all of
   * them, or -- option filter-two-pass -- the rows behind the early pass's share */
  uint64_t last_filter_final_rows;
} vk_index_stats;

/* ---- life cycle ------------------------------------------------------------------
 * BruteforceSearch(space, maxElements) bruteforce.h:54-64 /
 * HierarchicalNSW(space, max_elements, M, ef_construction, seed) hnswalg.h:121-179,
 * as called from VectorFlat::Create vector_flat.cc:53-74 and VectorHNSW::Create
 * vector_hnsw.cc:84-108. */
int vk_index_create(const vk_index_params *params, vk_index **out);
void vk_index_destroy(vk_index *ix);
int vk_device_count(void);
const char *vk_last_error(void);
/* What the LOADED library thinks the two structs of this header measure (which = 0: vk_index_params, 1: vk_index_stats, else
 * 0): a binding compares it with its own sizeof before the first call -- vk_index_get_stats writes that many bytes. */
uint64_t vk_abi_struct_size(int which);

/* ---- mutations (writer phase) -------------------------------------------------------
 * addPoint: bruteforce.h:66-83 / hnswalg.h:1278-1340 (same label again = in-place update,
 * which is how ModifyRecordImpl is expressed: vector_flat.cc:178-193, vector_hnsw.cc:273-286). */
int vk_index_add(vk_index *ix, uint64_t label, const void *row);
/* n rows, row-contiguous.  HNSW inserts them with params.build_threads host threads
 * (the reference's writer pool calling addPoint concurrently: valkey_search.cc:1171-1174), or on the device.
 * The rows take effect in their order: a label twice in the batch = the later row stands; at the capacity limit the
 * call returns VK_ERR_CAPACITY with every row in front of the one addPoint would have thrown at in the index and none
 * behind it.  Any other failure (VK_ERR_INTERNAL: the device had no room) may leave some of the batch's rows in; the
 * batch may be sent again as it is. */
int vk_index_add_batch(vk_index *ix, const uint64_t *labels, const void *rows, uint64_t n);
/* removePoint bruteforce.h:92-113 (last element moves into the hole) /
 * markDelete hnswalg.h:1173-1187 (tombstone). */
int vk_index_remove(vk_index *ix, uint64_t label);
/* resizeIndex: bruteforce.h:209-211 / hnswalg.h:758-777 (ResizeIfFull: vector_flat.cc:137-155,
 * vector_hnsw.cc:238-271 grow by block_size when add returns VK_ERR_CAPACITY). */
int vk_index_resize(vk_index *ix, uint64_t new_max_elements);
/* HierarchicalNSW::setEf hnswalg.h:210 */
int vk_index_set_ef(vk_index *ix, uint32_t ef);
/* publish staged mutations to HBM; call at the write->read phase switch */
int vk_index_flush(vk_index *ix);

/* ---- queries (reader phase) ----------------------------------------------------------
 * searchKnn: bruteforce.h:116-145 / hnswalg.h:1659-1725.
 *   k            : FLAT clamps to the element count like vector_flat.cc:234-236
 *   ef_runtime   : HNSW per-query EF_RUNTIME, 0 = index default (std::nullopt)
 *   allow_bits   : optional filter, the materialised BaseFilterFunctor (hnswlib.h:144-149):
 *                  bit `label` set = allowed, labels >= allow_nbits rejected; NULL = no filter
 *   cancel_flag  : optional host word, non-zero = cancelled (BaseCancellationFunctor,
 *                  hnswlib.h:153-157).  The calling thread watches it while it waits for the
 *                  device and relays it to the running kernels, which poll between row tiles
 *                  (FLAT: bruteforce.h:129) / every few expanded nodes (HNSW: hnswalg.h:400-402)
 *                  and stop: the call returns within about a millisecond of the flag.  FLAT
 *                  returns what it has; HNSW returns what its result lists hold when partial_ok,
 *                  else VK_ERR_CANCELLED (vector_hnsw.cc:327-329)
 *   out_dist/out_label : caller buffers of k entries; *out_n receives the count */
int vk_index_search(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime,
                    const uint64_t *allow_bits, uint64_t allow_nbits,
                    const volatile int *cancel_flag, int partial_ok,
                    float *out_dist, uint64_t *out_label, uint64_t *out_n);
/* nq independent queries answered by one device pass (what a query coalescer between
 * the reader pool, search.cc:886-910, and the device submits).  Outputs are [nq][k]
 * with out_n[nq]; the same filter applies to every query of the batch. */
int vk_index_search_batch(vk_index *ix, const void *queries, uint64_t nq, uint64_t k,
                          uint64_t ef_runtime, const uint64_t *allow_bits, uint64_t allow_nbits,
                          const volatile int *cancel_flag, int partial_ok,
                          float *out_dist, uint64_t *out_label, uint64_t *out_n);
/* The same with ONE FILTER PER QUERY: allow_bits_tab[q] / allow_nbits_tab[q] is query q's bitmap (NULL = unfiltered).
 * The reference evaluates its InlineVectorFilter per FT.SEARCH (search.cc:103-134), so a batch of coalesced hybrid
 * queries carries as many filters as queries; HNSW applies them inside one launch, FLAT (always pre-filtered by the
 * planner, planner.cc:23-29) serves the batch in runs of queries that share a bitmap. */
int vk_index_search_batch_filters(vk_index *ix, const void *queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                                  const uint64_t *const *allow_bits_tab, const uint64_t *allow_nbits_tab,
                                  const volatile int *cancel_flag, int partial_ok,
                                  float *out_dist, uint64_t *out_label, uint64_t *out_n);
/* Same, but queries and outputs are DEVICE pointers and the work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = the index's own stream) without a host sync:
 * what a sharded index calls on each of its shards (on a sharded index itself the pointers are
 * on the first shard's device and the call covers broadcast, shard searches, gather and merge).
 * Entries past the count are filled with (+inf, UINT64_MAX); a shard with fewer than k rows (or none)
 * answers with what it has.  Concurrent calls are safe (each takes its own scratch context; a context's
 * next user is ordered behind the work still in flight).  The caller must synchronise the stream before
 * the next writer phase (vk_index_flush / add / remove publish into the arrays the kernels read). */
int vk_index_search_batch_device(vk_index *ix, const void *d_queries, uint64_t nq, uint64_t k,
                                 uint64_t ef_runtime, const uint64_t *d_allow_bits,
                                 uint64_t allow_nbits, float *d_out_dist, uint64_t *d_out_label,
                                 uint32_t *d_out_n, void *hip_stream);
/* Exact kNN over an explicit label list: the pre-filter path
 * (search.cc:457-481 CalcBestMatchingPrefilteredKeys -> vector_base.cc:509-530
 * AddPrefilteredKey): heap of k, a later key replaces the top only on strictly
 * smaller distance.  Unknown / tombstoned labels are skipped. */
int vk_index_search_labels(vk_index *ix, const void *query, uint64_t k, const uint64_t *labels,
                           uint64_t n_labels, float *out_dist, uint64_t *out_label,
                           uint64_t *out_n);
/* fstdistfunc_(query, stored row) for one record
 * (ComputeDistanceFromRecordImpl: vector_flat.cc:256-271, vector_hnsw.cc:369-383) */
int vk_index_distance(vk_index *ix, uint64_t label, const void *query, float *out);
/* the stored row (GetValueImpl -> getPoint bruteforce.h:85-90 / getDataByInternalId) */
/* Query coalescing for vk_index_search.  The reference issues ONE query per call from up to
 * `reader-threads` pool threads (search.cc:886-910 -> :135-170); with max_batch > 1, concurrent
 * vk_index_search calls that agree on (k, ef_runtime) are merged into one device batch -- each with
 * its own filter bitmap or none (one filter per query inside the batch) and its own cancellation flag
 * (a caller whose flag is raised while it waits leaves at once; the batch itself runs to its end): the
 * first caller waits until max_batch calls are queued, max_wait_us elapsed, or nobody has arrived for a
 * quarter of max_wait_us (20-200 us: a lone caller is not held for the whole window, and the callers of a
 * batch that just finished -- who come back within microseconds of each other -- are all taken along), runs
 * the batch and hands each caller its own answer (identical to the answer it would have got alone).
 * max_batch <= 1 turns it off (the default). */
int vk_index_set_coalescing(vk_index *ix, uint32_t max_batch, uint32_t max_wait_us);
/* NON-BLOCKING single-query search: the shape of query::SearchAsync (src/query/search.cc:886-910), which queues the
 * request (up to max-query-queue-depth = 100 000 of them, src/valkey_search_options.cc:231-234) and continues on the
 * main thread when it completes.  The request joins the same (k, ef_runtime) lanes as coalesced vk_index_search calls;
 * the library keeps `batches-in-flight` device batches going (option, default 2: batch N+1 is collected, uploaded and
 * enqueued while batch N runs), so the number of queries in flight is bounded by the queue depth, not by the number
 * of reader threads.  Coalescing must be on (vk_index_set_coalescing with max_batch > 1), else VK_ERR_INVALID.
 *   done(user, status) is called exactly once, from a library thread, after out_dist / out_label / *out_n have been
 *   written; status is the vk_status of the batch the request travelled in (VK_ERR_CANCELLED as for vk_index_search).
 *   The callbacks of a batch run on the library's completer threads (option completer-threads, default 6; one pool for all
 *   indexes of the process), several
 *   side by side and in no particular order; one must not block for long (it holds up the completions queued behind
 *   it) and must not destroy the index.
 *   The query is copied at submission; allow_bits, cancel_flag and the output buffers must stay valid until then.
 * Returns VK_OK when the request was queued (the callback WILL fire), VK_ERR_BUSY when the queue is full (it will not).
 * vk_index_destroy answers what is still queued and waits for the callbacks; do not submit concurrently with it. */
typedef void (*vk_search_done_fn)(void *user, int status);
int vk_index_search_submit(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime,
                           const uint64_t *allow_bits, uint64_t allow_nbits,
                           const volatile int *cancel_flag, int partial_ok,
                           float *out_dist, uint64_t *out_label, uint64_t *out_n,
                           vk_search_done_fn done, void *user);
/* Completions told in bulk.  The reference's completion (search.cc:905-908) runs once per FT.SEARCH and re-posts to the main
 * thread; at half a million HNSW queries per second per GPU that per-request hand-over -- a callback, a queue operation and a
 * wake each -- is what bounds the serving rate (r05: 7 us per callback on four completer threads).  With a hook set, the
 * completions of EVERY request submitted to this index are told through it: one call per piece of a finished batch (option
 * handout-chunk; a request answered alone -- its token went up, the index is shutting down -- is a call with n = 1), from a
 * completer thread, after the members' outputs have been written.  items[i].user is the `user` given at submission, .status
 * what its callback would have received; the per-request callbacks are then NOT called (vk_index_search_submit still needs a
 * non-NULL one).  The caller answers the span in one go: one queue operation and one wake towards its main thread per piece.
 * Set it before the first submission (NULL restores per-request callbacks); the rules for a callback apply to the hook. */
typedef struct vk_completion {
  void *user;
  int32_t status;   /* vk_status */
  uint32_t reserved;
} vk_completion;
typedef void (*vk_batch_done_fn)(void *hook_user, const vk_completion *items, uint64_t n);
int vk_index_set_batch_completion(vk_index *ix, vk_batch_done_fn hook, void *hook_user);

/* ---- device-resident filters ---------------------------------------------------------------------------------------------
 * The reference filters an HNSW search with InlineVectorFilter (src/query/search.cc:103-134), a functor called per visited
 * candidate.  A kernel cannot call a functor: a search carries a bitmap over the labels.  What the query layer HOLDS for a
 * predicate is not a bitmap but the EntriesFetchers of its terms (search.cc:301-399; src/indexes/tag.cc:383-455 yields the
 * keys of the matched tags) -- lists of keys, i.e. of internal ids.  vk_filter_create builds the bitmap ON THE DEVICE from
 * such lists: `labels` in any order, duplicates allowed (the union of several fetchers, search.cc:208-220), and / or sorted
 * `runs` of consecutive ids ([n_runs][2] = first, last inclusive; what a numeric range over ids assigned in ingest order
 * yields), on top of an optional host bitmap `base_bits` of nbits bits (NULL = start empty).  Labels >= nbits are ignored
 * (rejected by every search, like allow_nbits above).  Cost: 8 B per id over PCIe + one scatter pass, instead of a host sweep
 * over every label of the index per query (r04: 10M functor calls per FT.SEARCH).
 * A filter is reference counted: the creator holds one reference (vk_filter_release drops it), every search in flight holds
 * one, so a caller may release -- or time out and leave -- while its batch is still on the device.  A filter belongs to
 * the index it was created for (its bitmap lives on that index's device(s)) and is immutable.
 *   vk_filter_combine: a OP b (0 = and, 1 = or, 2 = a and not b) of two filters of one index and one nbits, on the device:
 *   composed predicates over cached terms (search.cc:326-360 intersects by re-evaluating the predicate per key). */
typedef struct vk_filter vk_filter;
int vk_filter_create(vk_index *ix, uint64_t nbits, const uint64_t *labels, uint64_t n_labels, const uint64_t *runs,
                     uint64_t n_runs, const uint64_t *base_bits, vk_filter **out);
int vk_filter_combine(vk_index *ix, const vk_filter *a, const vk_filter *b, uint32_t op, vk_filter **out);
/* ... and n of them at once (a batch of requests each with its own composed predicate): out[i] = a[i] ops[i] b[i], one launch
 * and one wait per device instead of n (n <= 65535; all or nothing: on an error no out[i] is set). */
int vk_filter_combine_batch(vk_index *ix, const vk_filter *const *a, const vk_filter *const *b, const uint32_t *ops, uint64_t n,
                            vk_filter **out);
void vk_filter_retain(vk_filter *f);
void vk_filter_release(vk_filter *f);
/* nbits, and the number of allowed labels (counted on the device: what query::UsePreFiltering, planner.cc:21-45, wants as
 * its estimate) */
int vk_filter_info(const vk_filter *f, uint64_t *out_nbits, uint64_t *out_allowed);
/* the bitmap back on the host: n_words words, zero filled past the filter's end (tests, debugging) */
int vk_filter_read(const vk_filter *f, uint64_t *out_words, uint64_t n_words);
/* Cache: filters of one index under a caller-chosen key (e.g. the predicate's canonical text, "@tag:{x}") and EPOCH -- any
 * counter that changes whenever the answer of the predicate may have changed (the module: the count of write phases of the
 * schema's time-sliced mutex, src/index_schema.cc:285-292; finer: a mutation counter of the attribute's own index).  get
 * returns a retained handle in *out, or NULL when the key is unknown or was stored under another epoch (the stale entry is
 * dropped); put stores (and retains) the handle, replacing an older entry; the cache is bounded by the options
 * filter-cache-entries / filter-cache-bytes, least recently used first. */
int vk_index_filter_cache_get(vk_index *ix, const void *key, uint64_t key_len, uint64_t epoch, vk_filter **out);
int vk_index_filter_cache_put(vk_index *ix, const void *key, uint64_t key_len, uint64_t epoch, vk_filter *f);
/* searches with a filter handle (NULL = unfiltered): as vk_index_search / vk_index_search_submit / vk_index_search_batch_filters */
int vk_index_search_filter(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, vk_filter *filter,
                           const volatile int *cancel_flag, int partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n);
int vk_index_search_submit_filter(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, vk_filter *filter,
                                  const volatile int *cancel_flag, int partial_ok, float *out_dist, uint64_t *out_label,
                                  uint64_t *out_n, vk_search_done_fn done, void *user);
int vk_index_search_batch_filter_handles(vk_index *ix, const void *queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                                         vk_filter *const *filters, const volatile int *cancel_flag, int partial_ok,
                                         float *out_dist, uint64_t *out_label, uint64_t *out_n);

/* Run-time options: the analogue of `CONFIG SET search.<name>` (src/valkey_search_options.cc:74-81 hnsw-block-size,
 * :150-162 hnsw-allow-replace-deleted / hnsw-validation-enable, :231-234 max-query-queue-depth, :363-390
 * prefiltering-threshold-ratio).  Names (csrc/options.hpp has the table with defaults and ranges), e.g.
 *   coalesce-max-batch, coalesce-max-wait-us, max-query-queue-depth, batches-in-flight, shard-ef-pct, shard-gather,
 *   kernel-timing, flat-filter, filter-spill-chunks, hnsw-visited-bytes, hnsw-pool-bytes, ...
 * A sharded index applies an option to itself and to every shard.  Unknown name or value out of range: VK_ERR_INVALID.
 * Options take effect for searches that START after the call; no search path reads the environment. */
int vk_index_set_option(vk_index *ix, const char *name, uint64_t value);
int vk_index_get_option(vk_index *ix, const char *name, uint64_t *out_value);
int vk_index_get_row(vk_index *ix, uint64_t label, void *out_row);
int vk_index_contains(vk_index *ix, uint64_t label, int *out_found);
int vk_index_get_stats(vk_index *ix, vk_index_stats *out);

/* ---- device-resident bulk load (benchmarks / GPU-side ingest) --------------------------
 * Rows already in HBM in the index's own row layout (row i at d_rows + i*row_stride_bytes,
 * `dim` elements then zero padding): reserve, let the caller fill, then commit with labels
 * (NULL = 0..n-1).  FLAT only. */
int vk_index_device_rows(vk_index *ix, uint64_t n_rows, void **d_rows, uint64_t *row_stride_bytes);
int vk_index_commit_device_rows(vk_index *ix, uint64_t n_rows, const uint64_t *labels);
/* the same for ONE shard of a sharded index (rows on that shard's device; labels are required and must be unique
 * across the shards); vk_index_shard_count is 0 for a plain index */
int vk_index_shard_count(vk_index *ix, uint32_t *out_n);
/* the statistics of ONE shard (vk_index_get_stats of a sharded index sums the shards' cumulative counters; a per-shard
 * rate -- the slowest shard's kernel time per launch -- needs the shards' own deltas) */
int vk_index_shard_stats(vk_index *ix, uint32_t shard, vk_index_stats *out);
int vk_index_shard_device_rows(vk_index *ix, uint32_t shard, uint64_t n_rows, void **d_rows, uint64_t *row_stride_bytes);
int vk_index_shard_commit_device_rows(vk_index *ix, uint32_t shard, uint64_t n_rows, const uint64_t *labels);

/* ---- shard merge (multi-GPU) ---------------------------------------------------------------
 * k smallest by (distance,label) out of `parts` per-shard lists: the role of
 * SearchPartitionResultsTracker::AddResult (fanout.cc:162-175), with the total order of
 * bruteforce.h's heap so that an n-shard answer equals the 1-shard answer.
 * Device pointers; lists laid out [parts][nq][k]; enqueued on hip_stream. */
int vk_merge_topk_device(const float *d_dist, const uint64_t *d_label, uint32_t parts, uint64_t nq,
                         uint64_t k, float *d_out_dist, uint64_t *d_out_label, uint32_t *d_out_n,
                         int device_id, void *hip_stream);

/* ---- persistence: SaveIndex / LoadIndex chunk streams ----------------------------------------
 * bruteforce.h:147-207, hnswalg.h:808-1139 via RDBChunkOutputStream / RDBChunkInputStream
 * (rdb_serialization.h:289-340).  The callbacks carry one chunk each. */
typedef int (*vk_write_chunk_fn)(void *user, const void *data, uint64_t len);
typedef int (*vk_read_chunk_fn)(void *user, void *buf, uint64_t cap, uint64_t *len);
int vk_index_save(vk_index *ix, vk_write_chunk_fn write_chunk, void *user);
int vk_index_load(const vk_index_params *params, vk_read_chunk_fn read_chunk, void *user, vk_index **out);
/* The same with the VectorTracker hook of LoadIndex (bruteforce.h:171-207 at :201, hnswalg.h:887-1139 at :1000: every loaded vector is
 * handed to vector_tracker->TrackVector(label, data, len), which is how VectorBase gets its interned vectors back after a
 * restart, vector_base.cc:333-338): on_row(row_user, label, row) is called once per element with the dim f32 values as
 * they lie in the stream, in stream order, from the loading thread; non-zero aborts the load (VK_ERR_INTERNAL). */
typedef int (*vk_row_fn)(void *user, uint64_t label, const void *row);
int vk_index_load_tracked(const vk_index_params *params, vk_read_chunk_fn read_chunk, void *user, vk_row_fn on_row,
                          void *row_user, vk_index **out);

#ifdef __cplusplus
}
#endif
#endif /* VK_INDEX_H_ */
