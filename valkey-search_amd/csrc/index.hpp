// index.hpp -- the objects behind the C ABI (include/vk_index.h).
//
//   FlatIndex  stands in for hnswlib::BruteforceSearch<float>  (third_party/hnswlib/bruteforce.h)
//   HnswIndex  stands in for hnswlib::HierarchicalNSW<float>   (third_party/hnswlib/hnswalg.h)
// Host side keeps what the reference keeps on the host (label<->slot maps, the HNSW
// graph's authoritative copy); the device holds the row table (row_store.hpp) and, for
// HNSW, a fixed-stride mirror of the link lists, and answers the searches.
#pragma once
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "../../include/vk_index.h"
#include "filter_set.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "row_store.hpp"

namespace vk {

// grow-only device / pinned buffers
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  Status ensure(size_t bytes);
  void release();
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  Status ensure(size_t bytes);
  void release();
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// Results this small are written by the kernels straight into the pinned host buffers (hipHostMalloc memory is
// mapped into the device's address space): a one-at-a-time query pays no device-to-host copy calls, only the
// stream synchronisation.  Larger results go through device buffers and one copy each (PCIe bandwidth, not latency).
constexpr uint64_t kZeroCopyEntries = 4096;   // nq * k

// Per-call resources: a stream plus scratch, so that concurrent reader threads
// (search.cc:886-910 schedules one query per reader-pool thread) do not serialise.
struct SearchCtx {
  hipStream_t stream = nullptr;
  DevBuf d_q, d_part_d, d_part_l, d_out_d, d_out_l, d_out_n, d_allow, d_idx, d_tmp, d_stats, d_sync, d_pool, d_pool2, d_redo,
      d_fq16, d_fthr, d_fcnt, d_fcand, d_fspill, d_fsmax, d_fpart_d, d_fpart_l,   // candidate filter (flat_filter.hip)
      d_allow_tab;                                             // per-query filter table + the bitmaps behind it
  PinBuf h_q, h_out_d, h_out_l, h_out_n, h_tmp, h_idx, h_cancel;
  // per-member cancellation (SearchRequest::member_cancel): the words the kernel polls follow the batch word in h_cancel
  static constexpr size_t kMemberCancelOffset = 16;   // in u32 words
  Status wait(const volatile int *caller_flag, const volatile uint32_t *member_cancel, uint64_t nq);
  // In-kernel cancellation: the caller's flag (any host memory) cannot be read by the device, so the thread that
  // waits for the stream polls it and raises the context's own word in pinned memory, which the kernels poll
  // (FlatScanArgs::cancel).  arm_cancel: allocate + clear, returns the device-visible word (nullptr when the call
  // carries no flag); wait: hipStreamSynchronize, or with a flag a poll of stream and flag.
  Status arm_cancel(const volatile int *caller_flag, const uint32_t **device_word, uint64_t n_members = 0,
                    const uint32_t **member_words = nullptr);
  Status wait(const volatile int *caller_flag);
  // A device-buffer search (vk_index_search_batch_device) returns with its kernels still in flight on the CALLER's
  // stream and gives the context back: `busy` is recorded behind that work, and whoever leases the context next makes
  // its own stream wait for it before touching the scratch (begin_on) -- no host wait, and no two calls ever share
  // scratch that is live.  (DevBuf::ensure growing a buffer frees the old one with hipFree, which waits for the device.)
  hipEvent_t busy = nullptr;
  bool has_busy = false;
  // kernel-level timing of the candidate filter for vk_index_stats (bench.py's roofline): event pairs around its
  // launches, drained into the index's totals when a pair is reused or when the statistics are read
  struct TimedPair { hipEvent_t t0 = nullptr, t1 = nullptr; bool pending = false; };
  TimedPair timed[32];
  uint32_t timed_next = 0;
  Status begin_on(hipStream_t s);   // order the work about to be enqueued on `s` behind the context's previous user
  Status end_async(hipStream_t s);  // the work enqueued on `s` is the context's last user from now on
  ~SearchCtx();
};

class CtxPool {
 public:
  explicit CtxPool(int device, size_t max_ctx = 16) : device_(device), max_(max_ctx) {}
  ~CtxPool();
  // `on` = the stream the lease holder will enqueue on (nullptr = the context's own): it is ordered behind the
  // context's previous asynchronous user (SearchCtx::busy)
  SearchCtx *acquire(hipStream_t on = nullptr);
  void release(SearchCtx *c);
  // every context that is not leased right now
  template <class F> void for_each_free(F &&f) {
    std::lock_guard<std::mutex> lk(mu_);
    for (SearchCtx *c : free_) f(c);
  }

 private:
  int device_;
  size_t max_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<SearchCtx *> free_;
  std::vector<std::unique_ptr<SearchCtx>> all_;
};

struct CtxLease {
  CtxPool &pool;
  SearchCtx *ctx;
  explicit CtxLease(CtxPool &p, hipStream_t on = nullptr) : pool(p), ctx(p.acquire(on)) {}
  ~CtxLease() { pool.release(ctx); }
};

// The caller's cancellation word (BaseCancellationFunctor, hnswlib.h:153-157; the reference's cancel::Token is an atomic
// bool) is written by another thread while searches read it: a relaxed atomic load, not a plain read of a volatile int.
inline bool cancel_raised(const volatile int *flag) {
  return flag != nullptr && __atomic_load_n(const_cast<const int *>(flag), __ATOMIC_RELAXED) != 0;
}

struct SearchRequest {
  const float *queries = nullptr;   // host or device, [nq][dim] (host) / [nq][stride_f] padded (device)
  // host entry points only: one pointer per query instead of `queries` (the dispatcher's batches are made of single-query
  // calls whose vectors still lie in their callers' buffers; they are gathered straight into the pinned staging block)
  const float *const *query_tab = nullptr;
  uint64_t nq = 0, k = 0, ef = 0;
  const uint64_t *allow_bits = nullptr;
  uint64_t allow_nbits = 0;
  // one filter per query (host entry points: host arrays of host pointers, nullptr entry = unfiltered); overrides
  // allow_bits.  The reference builds its filter per FT.SEARCH (search.cc:103-134), so a batch of coalesced hybrid
  // queries carries as many bitmaps as queries.
  const uint64_t *const *allow_tab = nullptr;
  const uint64_t *allow_nbits_tab = nullptr;
  // device-resident filters (filter_set.hpp; host entry points): one for the whole batch, or one per query (a nullptr
  // entry = that query's allow_tab entry, or no filter).  They override allow_bits / allow_tab and are never uploaded.
  const FilterSet *filter = nullptr;
  const FilterSet *const *filter_tab = nullptr;
  const volatile int *cancel_flag = nullptr;
  bool partial_ok = true;
  // dispatcher batches: one host word per member, raised when THAT member's token goes up.  HNSW relays it to the wave
  // working on the member's query, which stops like the reference's loop does (hnswalg.h:400-402); FLAT ignores it (a
  // row pass costs the same with or without the member).
  const volatile uint32_t *member_cancel = nullptr;
  // search_device only: a device-visible cancellation word the caller maintains itself (the sharded index relays one
  // host flag to the kernels of every shard through it)
  const uint32_t *cancel_word = nullptr;
};

class Index {
 public:
  virtual ~Index() = default;
  const vk_index_params &params() const { return params_; }
  bool l2() const { return params_.metric == VK_METRIC_L2; }

  virtual Status add(uint64_t label, const float *row) = 0;
  virtual Status add_batch(const uint64_t *labels, const float *rows, uint64_t n) = 0;
  virtual Status remove(uint64_t label) = 0;
  virtual Status resize(uint64_t new_max) = 0;
  virtual Status set_ef(uint32_t ef) = 0;
  virtual Status flush() = 0;
  // host buffers in/out
  virtual Status search(const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) = 0;
  // device buffers in/out, enqueued on `stream` (nullptr = internal) without host sync
  virtual Status search_device(const SearchRequest &rq, float *d_out_dist, uint64_t *d_out_label,
                               uint32_t *d_out_n, hipStream_t stream) = 0;
  // distance of every listed label that the index holds (found[i] = 1), in list order: the device half of the
  // pre-filter path (ComputeDistanceFromRecordImpl per key, vector_base.cc:509-530)
  virtual Status label_distances(const float *query, const uint64_t *labels, uint64_t n, float *out_dist, uint8_t *found) = 0;
  // ... and its host half: AddPrefilteredKey's heap over those distances, in key order
  Status search_labels(const float *query, uint64_t k, const uint64_t *labels, uint64_t n, float *out_dist,
                       uint64_t *out_label, uint64_t *out_n);
  // sharded index: its shards (vk_index_shard_device_rows / _commit_device_rows address one of them)
  // the devices a filter of this index must be resident on (FilterSet::build)
  virtual void filter_devices(std::vector<int> *out) const = 0;
  virtual uint32_t shard_count() const { return 0; }
  virtual Status shard_device_rows(uint32_t, uint64_t, void **, uint64_t *) { return Status::Err(VK_ERR_INVALID, "not a sharded index"); }
  virtual Status shard_commit_device_rows(uint32_t, uint64_t, const uint64_t *) { return Status::Err(VK_ERR_INVALID, "not a sharded index"); }
  virtual Status shard_stats(uint32_t, vk_index_stats *) { return Status::Err(VK_ERR_INVALID, "not a sharded index"); }
  virtual Status distance(uint64_t label, const float *query, float *out) = 0;
  virtual Status get_row(uint64_t label, float *out) = 0;
  virtual Status contains(uint64_t label, bool *found) = 0;
  virtual Status stats(vk_index_stats *out) = 0;
  virtual Status device_rows(uint64_t n, void **d_rows, uint64_t *stride_bytes) = 0;
  virtual Status commit_device_rows(uint64_t n, const uint64_t *labels) = 0;
  virtual Status save(vk_write_chunk_fn fn, void *user) = 0;
  // run-time options (options.hpp): vk_index_set_option / vk_index_get_option; a sharded index forwards to its shards
  virtual Status set_option(const char *name, uint64_t value) { return opt_.set(name, value); }
  Status get_option(const char *name, uint64_t *out) const { return opt_.get(name, out); }
  const Options &options() const { return opt_; }

 protected:
  explicit Index(const vk_index_params &p) : params_(p) {
    if (p.shard_ef_pct) opt_.set(kOptShardEfPct, p.shard_ef_pct);
  }
  vk_index_params params_;
  Options opt_;
};

// vk_index_load_tracked: the VectorTracker hook of LoadIndex (bruteforce.h:201, hnswalg.h:1000) -- set by the ABI entry
// around the load on the loading thread, called by the loaders once per element as it is read from the stream
struct LoadObserver { vk_row_fn fn; void *user; };
extern thread_local const LoadObserver *g_load_observer;
inline Status observe_loaded_row(uint64_t label, const void *row) {
  if (g_load_observer && g_load_observer->fn(g_load_observer->user, label, row))
    return Status::Err(VK_ERR_INTERNAL, "load: the row callback failed");
  return Status::Ok();
}

Status create_sharded(const vk_index_params &p, std::unique_ptr<Index> *out);
Status load_sharded(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out);
Status create_flat(const vk_index_params &p, std::unique_ptr<Index> *out);
Status create_hnsw(const vk_index_params &p, std::unique_ptr<Index> *out);
Status load_flat(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out);
Status load_hnsw(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out);

// shared helpers --------------------------------------------------------------------
// pad nq host queries of `dim` floats into ctx->h_q ([nq][stride_f]) and copy to ctx->d_q
// (query_tab != nullptr: one host pointer per query instead of the contiguous block; parallel: large blocks copied by four threads)
Status upload_queries(SearchCtx *ctx, const float *queries, uint64_t nq, uint32_t dim, uint32_t stride_f, bool parallel = true,
                      const float *const *query_tab = nullptr);
Status upload_allow(SearchCtx *ctx, const uint64_t *allow_bits, uint64_t allow_nbits, const uint64_t **d_allow);
// an index without per-query filters in its kernels: the batch split into runs of queries that share a bitmap
Status search_grouped_by_filter(Index *ix, const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n);
// exact kNN over gathered distances with the AddPrefilteredKey rule (vector_base.cc:509-530)
void prefilter_heap_select(const float *dist, const uint64_t *labels, uint64_t n, uint64_t k,
                           float *out_dist, uint64_t *out_label, uint64_t *out_n);

// protobuf varint helpers for the index.proto headers (persist.cc)
void pb_put_varint_field(std::string &s, uint32_t field, uint64_t v);
void pb_put_double_field(std::string &s, uint32_t field, double v);
struct PbReader {
  const uint8_t *p, *end;
  bool next(uint32_t *field, uint32_t *wire, uint64_t *val);
};

}  // namespace vk
