// kernels.hpp -- host-visible launch interface of the HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// A/B constants of the launchers: fixed in the product library; the -DVK_EXPERIMENTS build (csrc/Makefile `experiments`)
// reads them from the environment once so that scripts/ can sweep them.
#ifdef VK_EXPERIMENTS
#include <stdlib.h>
#define VK_TUNE(env, dflt) (getenv(env) ? atoll(getenv(env)) : (long long)(dflt))
#else
#define VK_TUNE(env, dflt) ((long long)(dflt))
#endif

namespace vk {

// Dynamic LDS above the 64 KB default needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel function -- a property
// of the FUNCTION (per device), not of a launch.  r04 set it per launch to that launch's size: two dispatcher runners serving
// different (k, ef) lanes could interleave "set 80 KB / set 50 KB / launch 80 KB" and fail the launch.  It is
// now raised ONCE per (function, device) to the whole 160 KB of a CU; what a launch may use is still what it asks for.
hipError_t ensure_max_lds(const void *fn);

// K3: FLAT scan over rows [row_begin,row_end) for nq queries.
struct FlatScanArgs {
  const void *rows;           // [cap][row_stride_f] f32 or bf16 elements, zero padded to a multiple of 64 elements
  const uint64_t *labels;     // [cap]
  const float *queries;       // [nq][q_stride_f], padded like rows
  const uint64_t *allow_bits; // optional bitmap by label
  uint64_t allow_nbits;
  const float *lb_dist;       // optional per-query EXCLUSIVE lower bound (dist,label): only entries
  const uint64_t *lb_label;   // strictly greater are eligible (k > 1024 is served in passes of <= 1024)
  float *part_dist;           // partial top-k lists: [nq][nrp][k] per block when k <= 64, else [nq][nrp*4][k] per wave
  uint64_t *part_label;
  uint32_t row_stride_f, q_stride_f;
  uint32_t chunks;            // row_stride_f / 16
  uint32_t row_begin, row_end;
  uint32_t nq, k;
  uint32_t nrp;               // row partitions (blocks along the rows), multiple of 8
  uint32_t nqg;               // query groups = ceil(nq / kQB); grid = nrp * nqg blocks
  // optional cancellation word in host-pinned memory (BaseCancellationFunctor, hnswlib.h:153-157; the reference
  // polls it per row, bruteforce.h:129): polled once per kCancelPollTiles row tiles, non-zero = stop scanning and
  // hand over what the lists hold
  const uint32_t *cancel;
  // re-rank mode (the exact stage behind flat_filter_kernel): the rows of query q are the first
  // min(cand_cnt[q], cand_cap) entries of cand_row[q][..] instead of [row_begin, row_end); one query per block
  const uint32_t *cand_cnt;
  const uint32_t *cand_row;
  uint32_t cand_cap;
  // ... continued, for the queries that have more survivors than cand_cap, in spill chunks of kSpillChunk entries:
  // entry cand_cap + j of query q is cand_spill[(cand_qchunk[q * kSpillPerQuery + j / kSpillChunk] - 2) * kSpillChunk +
  // j % kSpillChunk]; a query whose cand_ovf word is up lost survivors (spill exhausted) and is skipped here -- the
  // exact redo pass answers it
  const uint32_t *cand_qchunk;
  const uint32_t *cand_spill;
  const uint32_t *cand_ovf;
  // redo mode (the exact pass over the queries the candidate filter handed over): the launch serves *nq_dev <= nq
  // queries, compact query i being query q_index[i] of the batch (its row in `queries`); partial lists are indexed by i
  const uint32_t *q_index;
  const uint32_t *nq_dev;
  // device-side conditional launch (run_flag == nullptr: always runs): the kernel returns at once unless
  // *run_flag == run_if, or -- with run_hi != 0 -- unless run_if <= *run_flag <= run_hi
  const uint32_t *run_flag;
  uint32_t run_if, run_hi;
  // fused re-rank (flat_rerank_kernel): per query the number of its blocks that have written their partial list (zeroed by
  // flat_qprep_kernel); the block that arrives last merges them and writes the answer
  uint32_t *done_cnt;
  // second bound of the fused re-rank (cand_val != nullptr): the filter's approximate score of every survivor, next to its
  // row slot (cand_val[q][i] / cand_spill_val[...] parallel to cand_row / cand_spill), the query's error polynomial
  // (FlatFilterArgs::qcoef) and the per-tile row norms.  The k-th largest of (score - margin) over a query's survivors
  // bounds its k-th best exact score from below -- from the WHOLE index this time, not from the sample -- and only the
  // survivors whose (score + margin) reaches it get an exact distance: a few dozen of the few hundred.
  const float *cand_val;
  const float *cand_spill_val;
  const float4 *qcoef;
  const uint32_t *tile_norm;
  uint32_t *reranked;         // optional [nq]: rows per query that got an exact distance (statistics; zeroed by flat_qprep_kernel)
#ifdef VK_EXPERIMENTS
  // the experiments build only: [nq][16] timestamps (wall_clock64, 100 MHz) of the fused re-rank's stages, written by wave 0
  // of each query's first block (scripts/rerank_stamps.py through vk_exp_rerank_stamps)
  unsigned long long *stamps;
#endif
};
constexpr uint32_t kSpillChunk = 4096;      // entries per spill chunk of the candidate filter's survivor lists
constexpr uint32_t kSpillPerQuery = 32;     // chunks one query may take (131072 survivors beyond its private list)
constexpr uint32_t kFilterRedoMax = 8;      // up to this many handed-over queries are re-scanned on their own; more: the whole batch
constexpr uint32_t kCancelPollTiles = 16;
constexpr uint32_t kCancelPollHops = 16;

// K4: batched FLAT (inner-product space) on the matrix cores, fused per-lane top-k
struct FlatGemmArgs {
  const void *rows;           // f32 or (bf16 = 1) bf16 rows, row_stride_f ELEMENTS apart
  uint32_t bf16;
  const uint64_t *labels;
  const float *queries;       // [nq][q_stride_f] padded
  const uint64_t *allow_bits;
  uint64_t allow_nbits;
  float *part_dist;           // [nq][nrp][8][k] per-lane partial lists
  uint64_t *part_label;
  uint32_t row_stride_f, q_stride_f, chunks;
  uint32_t n_rows, nq, k;
  uint32_t nrp;               // row partitions, multiple of 8
  uint32_t nqt;               // query tiles of tile_q
  uint32_t tile_q;            // queries per block: 32, 24 or 16 (flat_gemm_tile_q)
  // lockstep window W among the nqt blocks that stream the same row panel through one XCD's L2
  // (0 = off): a wave starts row tile t only after every sharer has started tile t-W
  uint32_t lockstep;
  uint32_t *sync;             // [nrp][4 waves][32] progress words, zeroed before the launch
  // per-query pruning bound shared by every list of the launch: order-preserving key of the
  // smallest k-th-best distance any FULL list has reached (0xFF800000 = +inf before the launch)
  uint32_t *qbound;           // [nq]
  uint32_t prepass;           // 1: this launch is the pre-pass (same code, separate kernel name)
  const float *init_bound;    // optional [nq]: a valid upper bound of each query's k-th best distance (pre-pass)
  uint32_t contig;            // 1: a row partition owns a contiguous range of tiles, 0: tiles rp, rp+nrp, ...
  const uint32_t *cancel;     // optional, as FlatScanArgs::cancel (polled every kCancelPollTiles 128-row tiles)
  const uint32_t *run_flag;   // as FlatScanArgs::run_flag
  uint32_t run_if, run_hi;
};
// K4h (flat_filter.hip): candidate stage of the batched FLAT search on the f16 matrix cores
struct FlatFilterArgs {
  const void *rows;           // f32 or (bf16 = 1) bf16 rows, row_stride_f elements apart
  uint32_t bf16;
  uint32_t l2;                // squared Euclidean distance instead of 1 - dot
  const uint32_t *hn16;       // l2: per row |x|^2 / 2 as two f16 (hi | lo << 16), written by row_stats_kernel
  const uint64_t *labels;
  const uint64_t *allow_bits;
  uint64_t allow_nbits;
  const float *queries;       // [nq][q_stride_f] f32, padded (input of flat_qprep_kernel)
  uint32_t q_stride_f;
  uint32_t qbf16;             // bf16 rows, inner-product space: q16 holds bf16 fragments and the bf16 matrix-core instruction multiplies
  uint32_t bdma;              // final pass: the B operands go L2 -> LDS by DMA, a ring of three stages (VK_FILTER_BDMA=0: through registers)
  uint32_t dma;               // with qbf16, final pass: the rows go HBM -> LDS by DMA (VK_FILTER_DMA=0: through registers)
  void *q16;                  // [nqt][row_stride_f/16][64][8] f16 (qbf16: bf16): the queries in MFMA fragment order (written by qprep)
  // per query column (written by qprep): the error margin of an approximate score against a row of norm R as a
  // polynomial  E(R) = c2 R^2 + c1 R + c0  (x, y, z), and the column's state (w: 0 = live, 1 = closed -- a padding
  // column, or a query that cannot go through f16 and was handed to the exact pass)
  float4 *qcoef;
  // per query column: a lower bound of the k-th best exact score in accumulator space (written by the bound
  // selection from the sample's group maxima; -inf = no bound, the gate is open)
  float *qbound;
  // per 128-row tile (row_stats_kernel): the largest row NORM of the tile as f32 bits, rounded up; +inf for a tile with
  // a value the f16 pipe cannot carry (non-finite, beyond 32768, half norm beyond f16 for L2): every pair of such a
  // tile survives and is settled by the exact re-rank
  const uint32_t *tile_norm;
  uint32_t *cand_cnt;         // [nq] survivors per query (zeroed by qprep; may exceed what was stored)
  uint32_t *cand_row;         // [nq][cap] their row slots ...
  float *cand_val;            // [nq][cap] ... and their approximate scores (accumulator space), for the re-rank's second bound
  uint32_t cap;
  uint32_t *qchunk;           // [nq][kSpillPerQuery] ... continued in spill chunks (2 + chunk index, 0 = none, 1 = being claimed; zeroed by qprep)
  uint32_t *spill;            // [n_chunks][kSpillChunk]
  float *spill_val;           // [n_chunks][kSpillChunk] scores of the spilled survivors
  uint32_t *spill_next;       // [1] chunks handed out (zeroed by qprep)
  uint32_t n_chunks;
  uint32_t *redo_cnt;         // [1] length of the redo list the final merge builds (zeroed by qprep)
  uint32_t *ovf_q;            // [nq] raised for a query that lost survivors (or cannot go through f16): the exact pass answers it
  uint32_t *done_cnt;         // [nq] the fused re-rank's arrival counters (zeroed by qprep; nullptr = not used)
  uint32_t *rerank_cnt;       // [nq] rows per query the re-rank evaluated (zeroed by qprep; statistics)
  // sample pass (mode 1): instead of gating, every (group of 64 rows, query) writes a LOWER BOUND of the group's best
  // exact score -- its best approximate score minus the margin -- to smax[q * smax_ld + group]; the k-th largest
  // of a query's group bounds bounds its k-th best exact score from below (k distinct rows reach it)
  uint32_t mode;
  float *smax;
  uint32_t smax_ld;
  uint32_t smax_fine;         // 1: eight groups of 16 rows per tile and query instead of two of 64 (a small sample)
  // The sample is made of n_tiles "sample tiles" of 128 rows spread over the WHOLE index: a sample tile is 16 (bf16
  // rows: 8) runs of 8 (16) rows sample_gap apart, the runs n_tiles * 8 * sample_gap rows apart (sample_row() in
  // flat_filter.hip), so index neighbours beyond a run land in different sample tiles, hence in different groups -- a
  // narrow stretch of similar rows (an index loaded cluster by cluster) still puts k of its rows into k distinct groups.  Witness rows carry the margin of the norm cap *norm_cap (row_stats: a robust upper norm
  // of the index's tiles); a row from a tile beyond the cap is no witness (the producers poison it with a NaN, which
  // the group maximum ignores).  qwit[j] = the column's margin at the cap (qprep).
  uint32_t sample_gap;
  const uint32_t *norm_cap;
  float *qwit;
  uint32_t n_tiles;           // tiles this launch walks (mode 0: ceil(n_rows / 128); mode 1: sample tiles)
  // A batch may walk the index in TWO launches of mode 0 (FlatIndex::scan_filter: the early pass and the main pass).  A block
  // owns the same contiguous range of the tile sequence in both; the early pass takes the first part_tiles tiles of every
  // range -- one stretch per block, spread evenly over the whole index -- and the main pass starts behind them
  // (part_first).  part_tiles = 0: the whole range in one launch.  early = 1 selects the early pass's kernel symbol (the
  // same code under another name, so that a profile tells the two apart).
  uint32_t part_first, part_tiles, early;
  uint32_t row_stride_f, n_rows, nq;
  uint32_t nqt;               // query tiles of 32 (<= 8 per launch)
  const uint32_t *cancel;
#ifdef VK_EXPERIMENTS
  // the -DVK_EXPERIMENTS build only (csrc/Makefile `experiments`): kernels whose answers are INVALID, for timing
  uint32_t timing;            // VK_FILTER_TIMING=1: the kernel variant with cycle counters per phase (f32 rows, IP only)
  unsigned long long *dbg;    // timing: [9] cycles per phase, summed over the waves (see the kernel)
  uint32_t prio;              // wave priorities of the three roles, 2 bits each (rows | queries << 2 | consumers << 4)
  uint32_t ablate_on, ablate; // VK_FILTER_ABLATE: pieces of the pipeline switched off, see flat_filter_body
  // margin audit (tests/helpers/exp_margin_check.py through vk_exp_filter_dump): the final pass writes what its gate saw for the
  // first dump_rows rows -- the approximate score of every (row, query) pair and the threshold of every (tile, query) --
  // so that a test can hold them against exact arithmetic.  Answers stay valid.
  float *dump_scores;         // [dump_rows][dump_ld]
  float *dump_thr;            // [dump_rows / 128][dump_ld]
  uint32_t dump_rows, dump_ld;
#endif
};
// bound selection: qbound[q] = the k-th largest of smax[q][0 .. groups) (-inf when fewer than k are finite)
struct FlatBoundArgs {
  const float *smax;
  uint32_t smax_ld, groups, k, nq;
  float *qbound;
};
// the early pass's harvest: qbound[q] = max(qbound[q], the k-th largest of (score - margin) over the query's survivors so far)
// -- k distinct rows reach it, so it bounds the k-th best exact score from below like the sample's bound does
// (flat_rerank_kernel takes its second bound the same way)
struct FlatTightenArgs {
  const uint32_t *cand_cnt;   // [nq]
  const uint32_t *cand_row;   // [nq][cap]
  const float *cand_val;      // [nq][cap]
  uint32_t cap, k, nq, l2;
  const float4 *qcoef;
  const uint32_t *tile_norm;
  float *qbound;
};
hipError_t launch_flat_bound_tighten(const FlatTightenArgs &a, hipStream_t s);
size_t flat_filter_lds_bytes();
bool flat_filter_supported(uint32_t row_stride_f, uint64_t k, bool bf16, bool l2);
// stats: [0] largest |row|^2 (SQUARED, reported only), [1] largest |element| of the index (f32 bits), [2] tiles flagged +inf
// so far, [3] the norm cap of the sample's witnesses: the NORM |row| (f32 bits, same unit as tile_norm) that 97 % of the finite
// tiles stay below (recomputed over n_tiles tiles)
hipError_t launch_row_stats(const void *rows, bool bf16, bool l2, uint32_t stride_e, uint32_t lo, uint32_t hi, uint32_t n_tiles,
                            uint32_t *stats, uint32_t *tile_norm, uint32_t *hn16, hipStream_t s);
hipError_t launch_flat_bound_select(const FlatBoundArgs &a, hipStream_t s);
constexpr uint32_t kFilterMaxGroups = 16384;   // group bounds per query the selection holds in registers (8192 sample tiles)
hipError_t launch_flat_qprep(const FlatFilterArgs &a, hipStream_t s);
hipError_t launch_flat_filter(const FlatFilterArgs &a, uint32_t blocks, hipStream_t s);

size_t flat_gemm_lds_bytes(uint32_t row_stride_f, uint32_t tile_q);
uint32_t flat_gemm_tile_q(uint32_t row_stride_f);
bool flat_gemm_supported(uint32_t row_stride_f, uint64_t k);
hipError_t launch_flat_gemm(const FlatGemmArgs &a, hipStream_t s);

struct MergeArgs {
  const float *in_dist;       // entry (part, q, i) at part*part_stride + q*q_stride + i
  const uint64_t *in_label;
  uint64_t part_stride, q_stride;
  uint32_t parts, per_part;
  uint32_t k;
  uint32_t out_ld;            // entries per query in the output arrays, 0 = k; entries past the count are (+inf, kNoLabel)
  float *out_dist;            // [nq][out_ld] ascending by (dist,label)
  uint64_t *out_label;
  uint32_t *out_n;            // [nq]
  const uint32_t *run_flag;   // as FlatScanArgs::run_flag
  uint32_t run_if, run_hi;
  // candidate-filter bookkeeping (optional): the block of a query whose ovf_q word is up appends it to redo_list
  // (redo_cnt = its length) and writes nothing -- the exact redo pass owns that query's output
  const uint32_t *ovf_q;
  uint32_t *redo_cnt;
  uint32_t *redo_list;
  // redo mode (as FlatScanArgs): block i serves compact query i < *nq_dev, reading lists at index i and writing the
  // output of query q_index[i]
  const uint32_t *q_index;
  const uint32_t *nq_dev;
};

struct GatherArgs {
  const void *rows;
  const float *query;         // padded
  const uint32_t *idx;        // [n] row slots
  float *out;                 // [n]
  uint32_t row_stride_f, chunks, n;
};

// K5 + K6: HNSW search, one wave per query (hnsw_search.hip)
struct HnswSearchArgs {
  const void *rows;            // [cap][row_stride_f] f32 or bf16 elements
  const uint64_t *labels;      // [cap]
  const uint32_t *links0;      // [cap][l0_stride]: word0 = count | tombstone<<16, then neighbour ids
  const uint32_t *upper_slot;  // [cap] first slot of the node's upper lists, 0xFFFFFFFF = level 0 only
  const uint32_t *upper_pool;  // [slots][up_stride]: slot (upper_slot[id] + level-1) = list at `level`
  const float *queries;        // [nq][q_stride_f] padded
  const uint64_t *allow_bits;  // optional, by label
  uint64_t allow_nbits;
  const uint64_t *const *allow_tab;   // optional [nq]: one bitmap per query (nullptr entry = no filter); overrides allow_bits
  const uint64_t *allow_nbits_tab;    // [nq]
  uint32_t *visited;           // [wave slots][bitmap_words] scratch: visited bitmaps, or (vis_hash_log2 != 0) hash sets
  float *out_dist;             // [nq][k]
  uint64_t *out_label;
  uint32_t *out_n;
  unsigned long long *stats;   // [5]: n_eval, n_hops, frontier entries dropped (must stay 0), queries, queries re-run (redo_in)
  unsigned long long *totals;  // optional [2]: n_eval, n_hops added to the INDEX's device-resident running totals as well
  uint32_t *queue;             // zeroed per launch: queries past the first wave of slots are taken in arrival order
  uint32_t row_stride_f, q_stride_f, chunks;
  uint32_t l0_stride, up_stride;
  uint32_t entry_point;
  int32_t max_level;
  uint32_t n_nodes, bitmap_words;
  uint32_t nq, k, ef;
  uint32_t cand_cap;           // frontier (candidate pool) entries per wave
  // Frontier in HBM instead of LDS (pool_g != nullptr): with a filter or tombstones the result list fills slowly and
  // the reference's candidate_set (an unbounded heap) grows to about ef / selectivity entries -- far more than fit
  // the LDS next to the query; per wave slot cand_cap distances followed by cand_cap ids
  float *pool_g;
  // 0 = frontier in LDS, 1 = in HBM with cand_cap <= 64k (a multiple of 128) and a query that outgrows it abandoned
  // into redo_out, 2 = in HBM with cand_cap >= n_nodes (a multiple of 8192; per wave slot also cand_cap / 64 segment
  // minima behind the ids): the launch that answers the abandoned queries, its work list is redo_in
  uint32_t gpool_level;
  uint32_t *redo_out;          // [1 + nq]: [0] = number of abandoned queries (zeroed before the launch), then their ids
  const uint32_t *redo_in;     // the same array, read by the second launch
  uint32_t nbr_cap;            // >= maxM0
  uint32_t check_deleted;      // any tombstones in the index
  uint32_t out_ids;            // 1: out_label receives internal ids (device-side graph construction)
  // optional cancellation word in host-pinned memory: polled at the start of a query and every kCancelPollHops
  // expanded nodes (the reference polls per popped candidate, hnswalg.h:400-402); a cancelled search keeps what its
  // result list holds, queries not started yet answer with empty lists
  const uint32_t *cancel;
  // optional, the same per QUERY ([nq] words in host-pinned memory): the members of a dispatcher batch carry their own
  // tokens; a raised word stops the wave that works on that query (or answers it empty if it has not started), the rest of
  // the batch runs on.  The reference stops one search within one distance evaluation (hnswalg.h:400-402)
  const uint32_t *cancel_q;
  // Visited set as an exact hash set of node ids (open addressing, 2^vis_hash_log2 words per wave slot = bitmap_words)
  // instead of one bit per node of the graph: a search touches a few thousand of 10M nodes, a table of 64 KB stays in
  // cache and is cleared in no time where the bitmap takes 1.25 MB per resident wave.  LDS-frontier launches only; a
  // query that would fill the table beyond 3/4 is abandoned into redo_out and re-run with the bitmap.
  uint32_t vis_hash_log2;
  // how the hash set is kept (option hnsw-visited-mode): 0 = compare-and-swap at agent scope (r02), 1 = the same at
  // WAVEFRONT scope -- the set is private to its wave, nothing outside it ever looks --, 2 = buckets in HBM with their
  // fill counts in LDS (no atomics on memory), 3 = the set in LDS (graphs below 2^24 nodes, ef <= ~450, ef x maxM0 <= hnsw-lds-visited-work:
  // 12 KB per wave hold about 5500 ids; ids that find no room on chip spill into the table in memory); the option's 4 = 3 whenever
  // the set fits the LDS at all (tests)
  uint32_t vis_mode;
};
// vis_mode 3: result lists in registers at up to eight slots per lane, two blocks of four waves per CU (ef up to ~450 at 768
// dimensions) and ef x maxM0 up to the option hnsw-lds-visited-work (14400) keep the visited set in 12 KB of LDS (ids beyond its
// ~5500 spill into the table in memory); hnsw-lds-visited-work-big (default 0 = off) takes a 32 KB set with one wave per SIMD
// (vis_mode 5, chosen by the host): measured slower than the small one + spill at every ef
constexpr int kHnswLdsList = 255;     // hnsw_slots_per_lane(): 1024 < ef <= kHnswMaxEf, result list in LDS (also 512 < ef <= 1024
                                      // when the frontier lives in HBM: those kernels have no 16-slot variant)
constexpr uint64_t kHnswMaxEf = 16384;   // (2 * ef words of LDS: the default max-vector-knn of 10000 fits, ft_search_parser.cc:34-45)
int hnsw_slots_per_lane(uint64_t ef);                      // 0 = ef beyond kHnswMaxEf
int hnsw_waves_per_block(const HnswSearchArgs &a);
bool hnsw_uses_latency_variant(const HnswSearchArgs &a);   // (a batch too small to fill the device)
size_t hnsw_lds_bytes(const HnswSearchArgs &a);
hipError_t hnsw_max_blocks(const HnswSearchArgs &a, bool l2, bool bf16, int e, int *blocks);
hipError_t launch_hnsw_search(const HnswSearchArgs &a, bool l2, bool bf16, int e, uint32_t blocks, hipStream_t s);

// K9 (hnsw_build.hip): level-0 neighbour selection and reverse links for a batch of new points
struct HnswBuildArgs {
  const void *rows;            // f32 or bf16 rows
  uint32_t row_stride_f, chunks;
  uint32_t *links0;            // [cap][l0_stride], updated in place
  uint32_t l0_stride;
  uint32_t max_keep;           // select: M; relink: maxM0
  // select: candidates of the new points first_id .. first_id + n_new - 1
  const uint64_t *cand_id;     // [n_new][cand_ld] ascending by distance (ids)
  const float *cand_dist;
  const uint32_t *cand_n;
  uint32_t cand_ld, n_new, first_id;
  uint32_t *sel_id;            // [n_new][max_keep]
  float *sel_dist;
  uint32_t *sel_n;
  // relink: CSR over the touched nodes
  const uint32_t *node;        // [n_touched]
  const uint32_t *off;         // [n_touched + 1]
  const uint32_t *add_p;       // new point ids, per node ascending by distance
  const float *add_d;
  uint32_t n_touched;          // number of touched nodes, or its upper bound when counts != nullptr
  const uint32_t *counts;      // optional device-side {touched nodes, pairs} (launch_hnsw_group)
};
// grouping of the selection output into that CSR on the device (radix sort by (node, distance))
struct HnswGroupArgs {
  const uint32_t *sel_id;      // [n_new][m]
  const float *sel_dist;
  const uint32_t *sel_n;
  uint32_t n_new, m, first_id;
  uint64_t *keys_a, *keys_b;   // [n_new * m] each
  uint32_t *vals_a;            // [n_new * m]
  uint32_t *flags, *pos;       // [n_new * m] each
  void *tmp;                   // hnsw_group_tmp_bytes(n_new * m)
  size_t tmp_bytes;
  uint32_t *node, *off;        // out: [n_new * m], [n_new * m + 1]
  uint32_t *add_p;             // out: [n_new * m]
  float *add_d;                // out: [n_new * m]
  uint32_t *counts;            // out: [2]
};
size_t hnsw_group_tmp_bytes(uint32_t n_pairs);
hipError_t launch_hnsw_group(const HnswGroupArgs &g, hipStream_t s);
hipError_t launch_hnsw_gather_lists(uint32_t *dst, const uint32_t *links0, uint32_t stride, uint32_t first, uint32_t n_new,
                                    const uint32_t *node, const uint32_t *counts, hipStream_t s);
size_t hnsw_build_lds_bytes(const HnswBuildArgs &a, bool relink);
hipError_t launch_hnsw_select(const HnswBuildArgs &a, bool l2, bool bf16, hipStream_t s);
hipError_t launch_hnsw_relink(const HnswBuildArgs &a, bool l2, bool bf16, hipStream_t s);
hipError_t launch_hnsw_widen_rows(const void *rows, uint32_t stride_e, uint32_t first, uint32_t n, float *out, hipStream_t s);

// scatter rows of u32 words: dst[idx[i]*stride + w] = src[i*stride + w]
hipError_t launch_scatter_u32(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride,
                              hipStream_t s);

// gather rows of u32 words: dst[i*stride + w] = src[idx[i]*stride + w]
hipError_t launch_gather_u32(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride,
                             hipStream_t s);

int flat_scan_slots_per_lane(uint64_t k);                 // 0 = k too large for the in-register top-k
int flat_scan_pick_qb(uint64_t nq, uint32_t chunks, int e, bool l2);
hipError_t launch_flat_scan(const FlatScanArgs &a, bool l2, bool bf16, int qb, int e, hipStream_t s);
hipError_t launch_merge_topk(const MergeArgs &a, int e, uint64_t nq, hipStream_t s);
hipError_t launch_gather_distance(const GatherArgs &a, bool l2, bool bf16, hipStream_t s);
// the answer of an empty index / shard: out_n = 0, every entry (+inf, kNoLabel)
// fused re-rank + selection of the candidate filter's survivors (k <= 64): one block per query writes the query's answer
hipError_t launch_flat_rerank(const FlatScanArgs &a, const MergeArgs &m, bool l2, bool bf16, hipStream_t s);
hipError_t launch_fill_empty(float *out_dist, uint64_t *out_label, uint32_t *out_n, uint32_t nq, uint32_t k, hipStream_t s);
// bound[q] = out_dist[q][k-1] if the query found k entries, +inf otherwise; bound[nq + q] = the same as an
// order-preserving u32 key (the buffer holds 2*nq words)
hipError_t launch_kth_bound(const float *out_dist, const uint32_t *out_n, uint32_t k, uint32_t nq, float *bound, hipStream_t s);

// filter_build.hip: allow-bitmaps built on the device from id lists / id runs (what the EntriesFetchers of a predicate
// yield, src/query/search.cc:301-399): bits must be zeroed (or hold the set to extend); labels >= nbits are ignored
// (d_partial[block] = bits the block turned on, filter_set_*_blocks() of them: the host adds them up)
uint32_t filter_set_ids_blocks(uint64_t n);
uint32_t filter_set_runs_blocks(uint64_t n_runs);
hipError_t launch_filter_set_ids(uint64_t *bits, uint64_t nbits, const uint64_t *d_ids, uint64_t n, unsigned long long *d_partial, hipStream_t s);
hipError_t launch_filter_set_runs(uint64_t *bits, uint64_t nbits, const uint64_t *d_runs, uint64_t n_runs, unsigned long long *d_partial,
                                  hipStream_t s);   // [n_runs][2] = first, last
hipError_t launch_filter_popcount(const uint64_t *bits, uint64_t words, unsigned long long *d_out, hipStream_t s);             // *d_out += set bits
// dst = a OP b, dst[words] = 0, d_partial[block] = set bits of the block's words (filter_combine_blocks(words) of them)
uint32_t filter_combine_blocks(uint64_t words);
hipError_t launch_filter_combine(uint64_t *dst, const uint64_t *a, const uint64_t *b, uint64_t words, uint32_t op, unsigned long long *d_partial,
                                 hipStream_t s);   // 0 and, 1 or, 2 and-not
// n combinations in one launch: d_items [n][4] = {dst, a, b, op} device pointers / op code, d_counts [n] zeroed (the results' bits)
hipError_t launch_filter_combine_batch(const uint64_t *d_items, uint32_t n, uint64_t words, unsigned long long *d_counts, hipStream_t s);

}  // namespace vk
