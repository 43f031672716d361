// vk_vector_adaptor.h -- the reference-side binding in the shape SURVEY section 8(b) asks for: classes DERIVED FROM
// valkey_search::indexes::VectorBase with the overrides of VectorFlat<T> / VectorHNSW<T>
// (src/indexes/vector_flat.h:37-63, vector_hnsw.h:36-73; the pure virtuals of vector_base.h:129-282), implemented on the
// C ABI of vk_index.h.  Source-only: in the module it is compiled with VK_ADAPTOR_IN_TREE defined (real headers); this
// repository compiles it against tests/helpers/mock_valkey_search.h with -Wall -Werror (tests/test_abi_symbols.py) and
// drives it on a GPU (tests/test_facade_gpu.py, scripts/adaptor_probe.cc), so a drift between these signatures and the
// interface is a build error.
//
// What differs from the hnswlib-backed classes, and why:
//   * SearchAsync() -- the entry the query layer should call (query::SearchAsync,
//     src/query/search.cc:886-910, schedules a task that RETURNS after the submission; the completion re-posts
//     MaybeAddIndexedContent + the callback to the pool).  The request is handed to vk_index_search_submit and the reader
//     thread is free again: queries in flight are bounded by max-query-queue-depth, not by reader-threads, which is what
//     lets single-query FT.SEARCH traffic fill device batches (r04's Search() parked the reader thread: 0.47 / 0.19 of the
//     device rate for FLAT / HNSW).  Search() -- same arguments as the reference's -- remains for callers that must block;
//     it goes through the library's blocking entry (one shared wake word per batch) and, on a FLAT index, keeps whole
//     batches together.
//   * Filters are DEVICE-RESIDENT allow-sets (vk_filter_*).  BuildFilter() walks the EntriesFetchers the query layer has
//     already built for the predicate (search.cc:301-399, :709) -- the keys that can match -- maps them to internal ids and
//     hands the id list to the device; nothing is evaluated for the labels the fetchers do not yield (r04 called the
//     functor for every label 0..max per query).  A filter is cached under the predicate's text and the adaptor's
//     write-phase epoch: a repeated `@tag:{x}` costs a hash lookup.  A BaseFilterFunctor passed to Search() is evaluated
//     over the TRACKED ids only.
//   * The cancellation token is time based and must be polled (src/utils/cancel.cc: IsCancelled checks the clock every
//     100th call).  Nobody waits inside an asynchronous search, so one watcher thread per process polls the tokens of the
//     requests in flight every 200 us and raises their cancel words; the library then answers the request at once and
//     stops the wave working on it (vk_index.h).  A request that carries its deadline is confirmed the moment it passes.
//   * GetValueImpl serves the tracked (interned) vector: the rows themselves live in HBM.
//   * LoadFromRDB() feeds the chunk iterator to vk_index_load_tracked; the rows come back through VectorBase::TrackVector
//     exactly as hnswlib's LoadIndex hands them to its VectorTracker (bruteforce.h:201, hnswalg.h:1000).
//   * RespondWithInfoImpl appends the counters metrics.h:40-50,75-80 and hnswalg.h:98-99,1199 expose
//     (vk_index_stats: searches, errors by kind, distance computations, hops, reclaimable bytes, latency histogram).
#ifndef VK_VECTOR_ADAPTOR_H_
#define VK_VECTOR_ADAPTOR_H_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <shared_mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#ifdef VK_ADAPTOR_IN_TREE
#include "absl/functional/any_invocable.h"
#include "src/attribute_data_type.h"
#include "src/indexes/vector_base.h"
#include "src/query/search.h"
#include "src/rdb_serialization.h"
#include "src/utils/cancel.h"
#endif
#include "vk_index.h"

namespace valkey_search::indexes {

#ifdef VK_ADAPTOR_IN_TREE
using VkChunkOut = RDBChunkOutputStream;       // (taken by value upstream: SaveIndexImpl(RDBChunkOutputStream chunked_out))
#define VK_CHUNK_OUT_PARAM RDBChunkOutputStream chunked_out
#else
#define VK_CHUNK_OUT_PARAM RDBChunkOutputStream &chunked_out
#endif

inline absl::Status VkToStatus(int rc) {
  switch (rc) {
    case VK_OK: return absl::OkStatus();
    case VK_ERR_INVALID: return absl::InvalidArgumentError(vk_last_error());
    case VK_ERR_NOT_FOUND: return absl::NotFoundError(vk_last_error());
    case VK_ERR_CANCELLED: return absl::CancelledError(query::kTimeoutMsg);          // vector_hnsw.cc:327-329
    case VK_ERR_BUSY: return absl::ResourceExhaustedError(vk_last_error());         // max-query-queue-depth reached
    case VK_ERR_NO_DEVICE: return absl::UnavailableError(vk_last_error());
    default: return absl::InternalError(vk_last_error());                           // what the hnswlib catch blocks return
  }
}
// (a status that arrives through a completion callback: the thread-local message of vk_last_error() belongs to another thread)
inline absl::Status VkCompletionStatus(int rc) {
  switch (rc) {
    case VK_OK: return absl::OkStatus();
    case VK_ERR_CANCELLED: return absl::CancelledError(query::kTimeoutMsg);
    case VK_ERR_INVALID: return absl::InvalidArgumentError("the vector backend rejected the search");
    case VK_ERR_BUSY: return absl::ResourceExhaustedError("the vector backend's query queue is full");
    default: return absl::InternalError("the vector backend failed the search");
  }
}

// A counted reference on a device-resident filter (vk_filter_*): copies retain, the destructor releases.  Empty = no filter.
class VkFilterRef {
 public:
  VkFilterRef() = default;
  static VkFilterRef Adopt(vk_filter *f) { VkFilterRef r; r.f_ = f; return r; }   // takes over the caller's reference
  VkFilterRef(const VkFilterRef &o) : f_(o.f_) { if (f_) vk_filter_retain(f_); }
  VkFilterRef(VkFilterRef &&o) noexcept : f_(o.f_) { o.f_ = nullptr; }
  VkFilterRef &operator=(VkFilterRef o) noexcept { std::swap(f_, o.f_); return *this; }
  ~VkFilterRef() { if (f_) vk_filter_release(f_); }
  vk_filter *get() const { return f_; }
  explicit operator bool() const { return f_ != nullptr; }
  uint64_t allowed() const { uint64_t n = 0; if (f_) (void)vk_filter_info(f_, nullptr, &n); return n; }

 private:
  vk_filter *f_ = nullptr;
};

// ---- the token watcher: one thread for every asynchronous search of the process ------------------------------------------
// cancel::Token::IsCancelled is made to be polled by ONE thread at a time and looks at the clock on every 100th call
// (cancel.cc: TimeoutPollFrequency).  While a request is registered here this thread is its only poller.
class VkTokenWatch {
 public:
  using Handle = uint32_t;   // slot << 3 | shard
  static VkTokenWatch &Instance() {
    static VkTokenWatch *w = new VkTokenWatch();   // (never destroyed: requests may complete during static destruction)
    return *w;
  }
  // (the slab is cut into kShards pieces with a lock each: one lock was taken twice per request by every reader thread and
  //  every completion thread, and by the watcher while it swept)
  Handle Register(cancel::Token token, volatile int *word, std::optional<std::chrono::steady_clock::time_point> deadline) {
    const uint32_t si = next_shard_.fetch_add(1, std::memory_order_relaxed) & (kShards - 1);
    Shard &sh = shards_[si];
    uint32_t slot;
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      if (!sh.free.empty()) {
        slot = sh.free.back();
        sh.free.pop_back();
      } else {
        slot = (uint32_t)sh.slots.size();
        sh.slots.emplace_back();
      }
      Slot &e = sh.slots[slot];
      e.token = std::move(token);
      e.word = word;
      e.has_deadline = deadline.has_value();
      if (deadline) e.deadline = *deadline;
      e.live = true;
    }
    // (the watcher is woken only out of its idle wait: a notify per registration made it sweep once per request)
    if (live_.fetch_add(1, std::memory_order_acq_rel) == 0) {
      std::lock_guard<std::mutex> lk(idle_mu_);
      if (!started_) {
        started_ = true;
        std::thread([this] { Loop(); }).detach();
      }
      idle_cv_.notify_one();
    }
    return (slot << 3) | si;
  }
  void Unregister(Handle h) {   // after this returns the watcher no longer touches the token or the word
    Shard &sh = shards_[h & (kShards - 1)];
    cancel::Token dropped;      // (released outside the lock)
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      Slot &e = sh.slots[h >> 3];
      e.live = false;
      dropped = std::move(e.token);
      e.word = nullptr;
      sh.free.push_back(h >> 3);
    }
    live_.fetch_sub(1, std::memory_order_acq_rel);
  }

 private:
  static constexpr uint32_t kShards = 8;
  struct Slot {
    cancel::Token token;
    volatile int *word = nullptr;                                       // raised once, never lowered
    std::chrono::steady_clock::time_point deadline{};                   // when known: the token is confirmed right there
    bool has_deadline = false, live = false;
  };
  struct Shard {
    std::mutex mu;
    std::vector<Slot> slots;
    std::vector<uint32_t> free;
    size_t cursor = 0;
  };
  static constexpr int kBurst = 128;            // > TimeoutPollFrequency: forces one look at the clock
  // A tick every 200 us looks at every request with a known deadline that has passed, and at 1/25 of the others (each token
  // is polled about every 5 ms: its own cadence is one look at the clock per 100 polls anyway).  A shard's slab is contiguous
  // and its lock is dropped every 256 slots: with 32 768 requests in flight the registering threads still get through.
  void Loop() {
    for (;;) {
      if (live_.load(std::memory_order_acquire) == 0) {
        std::unique_lock<std::mutex> lk(idle_mu_);
        idle_cv_.wait(lk, [&] { return live_.load(std::memory_order_acquire) != 0; });
      } else {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
      const auto now = std::chrono::steady_clock::now();
      for (Shard &sh : shards_) {
        std::unique_lock<std::mutex> lk(sh.mu);
        const size_t n = sh.slots.size();
        size_t slow_budget = std::max<size_t>(16, n / 25);
        for (size_t i = 0; i < n; ++i) {
          if ((i & 255) == 255) { lk.unlock(); lk.lock(); if (sh.slots.size() < n) break; }
          Slot &e = sh.slots[i];
          if (!e.live || *e.word) continue;
          bool up = false;
          if (e.has_deadline) {
            if (now < e.deadline) continue;
            for (int b = 0; b < kBurst && !up; ++b) up = e.token->IsCancelled();
          } else {
            // round robin over the tokens without a deadline
            if (slow_budget == 0 || (i < sh.cursor && sh.cursor < n)) continue;
            slow_budget -= 1;
            sh.cursor = i + 1;
            up = e.token->IsCancelled();
          }
          if (up) __atomic_store_n(const_cast<int *>(e.word), 1, __ATOMIC_RELAXED);
        }
        if (sh.cursor >= n || slow_budget != 0) sh.cursor = 0;
      }
    }
  }
  Shard shards_[kShards];
  std::atomic<uint32_t> next_shard_{0};
  std::atomic<size_t> live_{0};
  std::mutex idle_mu_;
  std::condition_variable idle_cv_;
  bool started_ = false;
};

template <typename T>
class VectorGpu : public VectorBase {
 public:
#ifdef VK_ADAPTOR_IN_TREE
  using SearchDone = absl::AnyInvocable<void(absl::StatusOr<std::vector<Neighbor>>)>;
#else
  using SearchDone = std::function<void(absl::StatusOr<std::vector<Neighbor>>)>;
#endif
  ~VectorGpu() override {
    if (owns_) vk_index_destroy(ix_);   // (answers what is queued and waits for the completions)
    else if (ix_) (void)vk_index_set_batch_completion(ix_, nullptr, nullptr);   // (an adopted index goes back as it came)
  }
  size_t GetDataTypeSize() const override { return sizeof(T); }
  size_t GetCapacity() const override { return Stats().capacity; }
  uint64_t GetMaxInternalLabel() const override { return max_label_.load(std::memory_order_relaxed); }
  size_t GetLabelCount() const override { return Stats().count; }
  int GetDimensions() const { return dimensions_; }
  vk_index *handle() const { return ix_; }
  const vk_index_params &params() const { return params_; }

  // ---- the write -> read phase switch (src/index_schema.cc:285-292): staged mutations are published to the device (single
  // AddRecord calls of the phase are linked in bulk, on the device when there are thousands) and filters cached before it
  // are no longer served
  absl::Status OnWritePhaseEnd() {
    filter_epoch_.fetch_add(1, std::memory_order_relaxed);
    return VkToStatus(vk_index_flush(ix_));
  }
  uint64_t FilterEpoch() const { return filter_epoch_.load(std::memory_order_relaxed); }

  // ---- filters --------------------------------------------------------------------------------------------------------
  // From the EntriesFetchers of the predicate (what DoSearchVector has in hand before it calls PerformVectorSearch,
  // search.cc:709: EvaluateFilterAsPrimary).  `matches(key)` is the per-key predicate evaluation of
  // EvaluatePrefilteredKeys (search.cc:401-455) for queries the fetchers alone do not solve (IsUnsolvedQuery, :196-206);
  // pass nullptr when they do.  `cache_key` (e.g. the filter's canonical text) makes the result reusable until the next
  // write phase.  Cost: one hash lookup per FETCHED key; nothing per label of the index.
  template <class Matches>
  absl::StatusOr<VkFilterRef> BuildFilter(std::queue<std::unique_ptr<EntriesFetcherBase>> &fetchers, Matches &&matches,
                                          absl::string_view cache_key = {}) {
    const uint64_t epoch = FilterEpoch();
    if (!cache_key.empty()) {
      vk_filter *hit = nullptr;
      if (int rc = vk_index_filter_cache_get(ix_, cache_key.data(), cache_key.size(), epoch, &hit); rc != VK_OK) return VkToStatus(rc);
      if (hit) return VkFilterRef::Adopt(hit);
    }
    std::vector<uint64_t> ids;
    while (!fetchers.empty()) {
      auto fetcher = std::move(fetchers.front());
      fetchers.pop();
      ids.reserve(ids.size() + fetcher->Size());
      for (auto it = fetcher->Begin(); !it->Done(); it->Next()) {
        const InternedStringPtr &key = **it;
        if constexpr (!std::is_same_v<std::decay_t<Matches>, std::nullptr_t>) {
          if (!matches(key)) continue;
        }
        auto id = this->GetInternalIdDuringSearch(key);       // (keys without a vector in this index are not candidates)
        if (id.ok()) ids.push_back(id.value());               // duplicates across fetchers are harmless: the bitmap is an OR
      }
    }
    return MakeFilter(ids, cache_key, epoch);
  }
  // From a functor (the reference's InlineVectorFilter, search.cc:103-134): evaluated once per TRACKED id
  absl::StatusOr<VkFilterRef> BuildFilter(hnswlib::BaseFilterFunctor &functor, absl::string_view cache_key = {}) {
    const uint64_t epoch = FilterEpoch();
    if (!cache_key.empty()) {
      vk_filter *hit = nullptr;
      if (int rc = vk_index_filter_cache_get(ix_, cache_key.data(), cache_key.size(), epoch, &hit); rc != VK_OK) return VkToStatus(rc);
      if (hit) return VkFilterRef::Adopt(hit);
    }
    std::vector<uint64_t> ids;
    {
      std::shared_lock<std::shared_mutex> l(tracked_mu_);
      ids.reserve(tracked_.size());
      for (const auto &kv : tracked_) ids.push_back(kv.first);
    }
    size_t kept = 0;
    for (uint64_t id : ids)
      if (functor(id)) ids[kept++] = id;
    ids.resize(kept);
    return MakeFilter(ids, cache_key, epoch);
  }

  // ---- searches -------------------------------------------------------------------------------------------------------
  // VectorFlat<T>::Search (vector_flat.cc:224-254) / VectorHNSW<T>::Search (vector_hnsw.cc:313-347) WITHOUT the wait: returns
  // once the request is queued in the library; `done` is called exactly once, from a library thread, with what Search()
  // would have returned (a non-OK return value means the request was not queued and `done` will not be called).
  absl::Status SearchAsync(absl::string_view query, uint64_t count, cancel::Token cancellation_token, VkFilterRef filter,
                           std::optional<size_t> ef_runtime, bool enable_partial_results, SearchDone done,
                           std::optional<std::chrono::steady_clock::time_point> deadline = std::nullopt) {
    if (query.size() != (size_t)GetVectorDataSize()) return absl::InvalidArgumentError("query vector of the wrong size");
    if (count == 0) { done(std::vector<Neighbor>{}); return absl::OkStatus(); }
    auto *rq = new AsyncSearch();
    rq->self = this;
    rq->done = std::move(done);
    rq->out.reset(new uint64_t[count + (count + 1) / 2]);   // labels, then distances: one allocation
    rq->dist = reinterpret_cast<float *>(rq->out.get() + count);
    rq->filter = std::move(filter);
    rq->cancel_word = cancellation_token && cancellation_token->IsCancelled() ? 1 : 0;
    rq->watched = static_cast<bool>(cancellation_token) && !rq->cancel_word;
    if (rq->watched) rq->watch = VkTokenWatch::Instance().Register(std::move(cancellation_token), &rq->cancel_word, deadline);
    std::vector<T> normalised;
    const void *q = NormalisedQuery(query, &normalised);   // (the library copies the query at submission)
    const int rc = vk_index_search_submit_filter(ix_, q, count, ef_runtime.value_or(0), rq->filter.get(), &rq->cancel_word,
                                                 enable_partial_results ? 1 : 0, rq->dist, rq->out.get(), &rq->n,
                                                 &AsyncSearch::Completed, rq);
    if (rc != VK_OK) {
      if (rq->watched) VkTokenWatch::Instance().Unregister(rq->watch);
      delete rq;
      return VkToStatus(rc);
    }
    return absl::OkStatus();
  }

  // Completions in bulk.  The library tells the completions of a piece of a finished batch in ONE call (vk_index.h:
  // vk_index_set_batch_completion); the replies of the piece are built here (CreateReply's key lookups) and handed on
  //   * one `done` call per request (the default: what query::SearchAsync's callback is today, search.cc:905-908), or
  //   * ALL AT ONCE to the sink set with SetBulkDone: the module posts ONE task to its main thread that answers the whole
  //     span (one queue operation and one wake per piece instead of one per FT.SEARCH -- at half a million HNSW queries a
  //     second the per-request hand-over, not the search, was the bound: r05 0.76-0.83 of the device rate).
  struct CompletedSearch {
    SearchDone done;
    absl::StatusOr<std::vector<Neighbor>> result;
  };
  using BulkDone = std::function<void(std::vector<CompletedSearch> &&)>;
  void SetBulkDone(BulkDone sink) { bulk_done_ = std::move(sink); }   // before the first SearchAsync

  // The blocking forms (the reference's signature, and the same with a prebuilt filter)
  absl::StatusOr<std::vector<Neighbor>> Search(absl::string_view query, uint64_t count, cancel::Token &cancellation_token,
                                               std::unique_ptr<hnswlib::BaseFilterFunctor> filter = nullptr,
                                               std::optional<size_t> ef_runtime = std::nullopt, bool enable_partial_results = false) {
    VkFilterRef f;
    if (filter) {
      auto built = BuildFilter(*filter);
      if (!built.ok()) return built.status();
      f = std::move(built.value());
    }
    return Search(query, count, cancellation_token, std::move(f), ef_runtime, enable_partial_results);
  }
  absl::StatusOr<std::vector<Neighbor>> Search(absl::string_view query, uint64_t count, cancel::Token &cancellation_token, VkFilterRef filter,
                                               std::optional<size_t> ef_runtime = std::nullopt, bool enable_partial_results = false) {
    if (query.size() != (size_t)GetVectorDataSize()) return absl::InvalidArgumentError("query vector of the wrong size");
    std::vector<T> normalised;
    const void *q = NormalisedQuery(query, &normalised);
    const uint64_t k = count;                                  // (FLAT clamps to the element count inside, vector_flat.cc:234-236)
    std::vector<float> dist(k ? k : 1);
    std::vector<uint64_t> label(k ? k : 1);
    uint64_t n = 0;
    // (a token that is already up travels with the request; after that the watcher polls it while this thread is parked)
    volatile int cancel_word = cancellation_token && cancellation_token->IsCancelled() ? 1 : 0;
    const bool watched = static_cast<bool>(cancellation_token) && !cancel_word;
    VkTokenWatch::Handle wh;
    if (watched) wh = VkTokenWatch::Instance().Register(cancellation_token, &cancel_word, std::nullopt);
    const int rc = vk_index_search_filter(ix_, q, k, ef_runtime.value_or(0), filter.get(), &cancel_word, enable_partial_results ? 1 : 0,
                                          dist.data(), label.data(), &n);
    if (watched) VkTokenWatch::Instance().Unregister(wh);
    // (HNSW without partial results answers a raised token with VK_ERR_CANCELLED -> CancelledError, vector_hnsw.cc:327-329;
    //  FLAT returns what its scan had, like bruteforce.h:125-129 -- VectorFlat::Search has no such branch)
    if (rc != VK_OK) return VkToStatus(rc);
    return Reply(dist.data(), label.data(), n);
  }

 protected:
  VectorGpu(IndexerType type, vk_algo algo, int dimensions, data_model::DistanceMetric metric, absl::string_view attribute_identifier,
            data_model::AttributeDataType attribute_data_type)
      : VectorBase(type, dimensions, attribute_data_type, attribute_identifier), algo_(algo) {
    distance_metric_ = metric;
    normalize_ = metric == data_model::DISTANCE_METRIC_COSINE;  // VectorBase::Init, vector_base.cc:140-150
  }
  vk_index_params MakeParams(vk_index_params p) const {
    p.struct_size = sizeof p;
    p.algo = algo_;
    p.dtype = VK_DTYPE_F32;
    p.dim = (uint32_t)dimensions_;
    p.metric = distance_metric_ == data_model::DISTANCE_METRIC_L2   ? VK_METRIC_L2
               : distance_metric_ == data_model::DISTANCE_METRIC_IP ? VK_METRIC_IP
                                                                    : VK_METRIC_COSINE;
    p.random_seed = 100;
    p.device_id = -1;
    return p;
  }
  absl::Status Serve(uint32_t reader_threads) {
    // Single-query calls become device batches; the submit path needs it on.  The batch is sized for the
    // DEVICE, not for the reader pool -- SearchAsync keeps more queries in flight than there are reader threads: a FLAT pass
    // costs the same for 1 query or 256 (and its window waits for the callers of the batch that has just finished), an HNSW
    // launch wants thousands of waves.  A lone query is held for a quarter of the window at most.
    const uint32_t device_batch = algo_ == VK_ALGO_FLAT ? 256u : 8192u;
    // completions arrive a piece of a batch at a time (BatchCompleted), not one call per request
    if (int rc = vk_index_set_batch_completion(ix_, &VectorGpu::BatchCompleted, this); rc != VK_OK) return VkToStatus(rc);
    return VkToStatus(vk_index_set_coalescing(ix_, std::max(reader_threads, device_batch), algo_ == VK_ALGO_FLAT ? 400 : 2000));
  }
  // an index that exists already (built by a bulk loader, or -- scripts/adaptor_probe.cc -- by the benchmark): served through
  // this object, destroyed by whoever made it.  The caller keeps it alive until every search through this object is done.
  // (a libvkindex.so of another round behind this header: vk_index_get_stats would write past -- or short of -- the struct)
  static absl::Status CheckAbi() {
    if (vk_abi_struct_size(0) != sizeof(vk_index_params) || vk_abi_struct_size(1) != sizeof(vk_index_stats))
      return absl::FailedPreconditionError("libvkindex.so was built from another vk_index.h than this adaptor");
    return absl::OkStatus();
  }
  absl::Status Adopt(vk_index *ix, vk_index_params p, uint32_t reader_threads) {
    if (auto st = CheckAbi(); !st.ok()) return st;
    params_ = MakeParams(p);
    ix_ = ix;
    owns_ = false;
    max_label_.store(Stats().max_label, std::memory_order_relaxed);
    return Serve(reader_threads);
  }
  absl::Status Open(vk_index_params p, uint32_t reader_threads) {
    if (auto st = CheckAbi(); !st.ok()) return st;
    params_ = MakeParams(p);
    if (int rc = vk_index_create(&params_, &ix_); rc != VK_OK) return VkToStatus(rc);
    return Serve(reader_threads);
  }
  // VectorFlat<T>::LoadFromRDB / VectorHNSW<T>::LoadFromRDB (vector_flat.cc:100-124, vector_hnsw.cc:135-166): the chunk
  // stream through vk_index_load_tracked; every loaded vector goes back to VectorBase::TrackVector like LoadIndex's
  // VectorTracker calls (bruteforce.h:201, hnswalg.h:1000); the id counter resumes behind the largest loaded label
  absl::Status OpenFromRDB(vk_index_params p, uint32_t reader_threads, SupplementalContentChunkIter &&iter) {
    if (auto st = CheckAbi(); !st.ok()) return st;
    params_ = MakeParams(p);
    struct Source {
      RDBChunkInputStream input;
      VectorGpu *self;
      absl::Status st;
    } src{RDBChunkInputStream(std::move(iter)), this, absl::OkStatus()};
    const int rc = vk_index_load_tracked(
        &params_,
        [](void *u, void *buf, uint64_t cap, uint64_t *len) -> int {
          auto *s = static_cast<Source *>(u);
          auto chunk = s->input.LoadChunk();
          if (!chunk.ok()) { s->st = chunk.status(); return 1; }
          if (chunk.value()->size() > cap) { s->st = absl::InternalError("RDB chunk larger than the index geometry allows"); return 2; }
          std::memcpy(buf, chunk.value()->data(), chunk.value()->size());
          *len = chunk.value()->size();
          return 0;
        },
        &src,
        [](void *u, uint64_t label, const void *row) -> int {
          auto *s = static_cast<Source *>(u);
          (void)s->self->VectorBase::TrackVector(label, const_cast<char *>(static_cast<const char *>(row)), (size_t)s->self->GetVectorDataSize());
          return 0;
        },
        &src, &ix_);
    if (rc != VK_OK) return src.st.ok() ? VkToStatus(rc) : src.st;
    max_label_.store(Stats().max_label, std::memory_order_relaxed);
    return Serve(reader_threads);
  }

  // AddRecordImpl with the resize-and-retry loop of vector_flat.cc:137-176 / vector_hnsw.cc:238-271.  The call STAGES the row
  // (vk_index.h): an HNSW index links what a write phase staged in bulk at OnWritePhaseEnd() -- on the device when there are
  // thousands (IndexSchema feeds one key at a time, src/index_schema.cc:755-791; r04 linked each on the host: 17 min for 10M)
  absl::Status AddRecordImpl(uint64_t internal_id, absl::string_view record) override {
    for (;;) {
      const int rc = vk_index_add(ix_, internal_id, record.data());
      if (rc == VK_ERR_CAPACITY) {
        const uint64_t grow = params_.block_size ? params_.block_size : 1024;
        if (int r2 = vk_index_resize(ix_, Stats().capacity + grow); r2 != VK_OK) return VkToStatus(r2);
        continue;
      }
      if (rc == VK_OK) {
        uint64_t cur = max_label_.load(std::memory_order_relaxed);
        while (internal_id > cur && !max_label_.compare_exchange_weak(cur, internal_id, std::memory_order_relaxed)) {}
      }
      return VkToStatus(rc);
    }
  }
  absl::Status RemoveRecordImpl(uint64_t internal_id) override { return VkToStatus(vk_index_remove(ix_, internal_id)); }
  // same label again = in-place update (vector_flat.cc:178-193 pokes data_; vector_hnsw.cc:273-286 markDelete + addPoint)
  absl::Status ModifyRecordImpl(uint64_t internal_id, absl::string_view record) override { return AddRecordImpl(internal_id, record); }

  absl::StatusOr<std::pair<float, hnswlib::labeltype>> ComputeDistanceFromRecordImpl(uint64_t internal_id,
                                                                                    absl::string_view query) const override {
    float d = 0;
    if (int rc = vk_index_distance(ix_, internal_id, query.data(), &d); rc != VK_OK) return VkToStatus(rc);
    return std::pair<float, hnswlib::labeltype>{d, (hnswlib::labeltype)internal_id};
  }
  absl::Status SaveIndexImpl(VK_CHUNK_OUT_PARAM) const override {
    struct Sink { RDBChunkOutputStream *out; absl::Status st; } sink{&chunked_out, absl::OkStatus()};
    const int rc = vk_index_save(
        ix_,
        [](void *u, const void *data, uint64_t len) -> int {
          auto *s = static_cast<Sink *>(u);
          s->st = s->out->SaveChunk(static_cast<const char *>(data), (size_t)len);
          return s->st.ok() ? 0 : 1;
        },
        &sink);
    if (!sink.st.ok()) return sink.st;
    return VkToStatus(rc);
  }
  // the interned vectors VectorBase hands over (vector_flat.cc:273-300): the rows themselves live in HBM
  void TrackVector(uint64_t internal_id, const InternedStringPtr &vector) override {
    std::unique_lock<std::shared_mutex> l(tracked_mu_);
    tracked_[internal_id] = vector;
  }
  bool IsVectorMatch(uint64_t internal_id, const InternedStringPtr &vector) override {
    std::shared_lock<std::shared_mutex> l(tracked_mu_);
    auto it = tracked_.find(internal_id);
    return it != tracked_.end() && it->second->Str() == vector->Str();
  }
  void UnTrackVector(uint64_t internal_id) override {
    std::unique_lock<std::shared_mutex> l(tracked_mu_);
    tracked_.erase(internal_id);
  }
  char *GetValueImpl(uint64_t internal_id) const override {
    std::shared_lock<std::shared_mutex> l(tracked_mu_);
    auto it = tracked_.find(internal_id);
    return it == tracked_.end() ? nullptr : const_cast<char *>(it->second->Str().data());
  }
  // the device-side counters behind FT.INFO / INFO SEARCH; returns the number of (name, value) PAIRS x 2 appended, like
  // the reference's RespondWithInfoImpl return values (vector_hnsw.cc:202-229)
  int RespondWithCounters(ValkeyModuleCtx *ctx) const {
    const vk_index_stats s = Stats();
    uint64_t errors = 0;
    for (int i = 1; i < VK_STATUS_COUNT; ++i) errors += s.search_errors[i];
    const std::pair<const char *, uint64_t> rows[] = {
        {"gpu_device_bytes", s.device_bytes},       {"gpu_searches", s.searches},
        {"gpu_search_errors", errors},              {"gpu_search_timeouts", s.search_errors[VK_ERR_CANCELLED]},
        {"gpu_distance_computations", s.total_n_eval}, {"gpu_hops", s.total_n_hops},
        {"gpu_reclaimable_bytes", s.tombstoned_bytes}, {"gpu_latency_sum_us", s.latency_sum_ns / 1000},
        {"gpu_queue_depth", s.queued_now},          {"gpu_rejected_busy", s.rejected},
        {"gpu_cancelled_early", s.cancelled_early}, {"gpu_filters_built", s.filters_built},
        {"gpu_filter_cache_hits", s.filter_cache_hits}, {"gpu_staged_adds", s.staged_adds},
        {"gpu_staged_adds_device", s.staged_adds_device},
        // the library's serving threads: device batches formed from single queries, and where the runners' / completers' time went
        {"gpu_batches", s.coalesced_batches},          {"gpu_batched_queries", s.coalesced_queries},
        {"gpu_runner_idle_us", s.dispatch_idle_us},    {"gpu_runner_search_us", s.dispatch_search_us},
        {"gpu_completer_us", s.dispatch_completer_us}};
    for (const auto &r : rows) {
      ValkeyModule_ReplyWithSimpleString(ctx, r.first);
      ValkeyModule_ReplyWithLongLong(ctx, (long long)r.second);
    }
    return 2 * (int)(sizeof rows / sizeof rows[0]);
  }
  vk_index_stats Stats() const {
    vk_index_stats s;
    std::memset(&s, 0, sizeof s);
    (void)vk_index_get_stats(ix_, &s);
    return s;
  }

  // CopyAndNormalizeEmbedding (vector_base.cc:112-124): sequential f32, 1 / magnitude, zero vector unchanged
  const void *NormalisedQuery(absl::string_view query, std::vector<T> *store) const {
    if (!normalize_) return query.data();
    const T *src = reinterpret_cast<const T *>(query.data());
    store->resize((size_t)dimensions_);
    T magnitude = 0;
    for (int i = 0; i < dimensions_; ++i) magnitude += src[i] * src[i];
    magnitude = std::sqrt(magnitude);
    const T norm = magnitude == (T)0 ? (T)1 : (T)1 / magnitude;
    for (int i = 0; i < dimensions_; ++i) (*store)[(size_t)i] = norm * src[i];
    return store->data();
  }
  // CreateReply, vector_base.cc:258-277: ascending by distance, labels whose key is gone are skipped
  std::vector<Neighbor> Reply(const float *dist, const uint64_t *label, uint64_t n) const {
    std::vector<Neighbor> out;
    out.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
      auto key = GetKeyDuringSearch(label[i]);
      if (!key.ok()) continue;
      // (moved, not copied: a key's reference count is one cache line every reply that names the key bounces between the
      //  completer threads -- the popular neighbours of an HNSW graph are in most replies)
      out.emplace_back(std::move(key.value()), dist[i]);
    }
    return out;
  }
  absl::StatusOr<VkFilterRef> MakeFilter(const std::vector<uint64_t> &ids, absl::string_view cache_key, uint64_t epoch) {
    vk_filter *f = nullptr;
    if (int rc = vk_filter_create(ix_, GetMaxInternalLabel() + 1, ids.data(), ids.size(), nullptr, 0, nullptr, &f); rc != VK_OK) return VkToStatus(rc);
    VkFilterRef ref = VkFilterRef::Adopt(f);
    if (!cache_key.empty()) (void)vk_index_filter_cache_put(ix_, cache_key.data(), cache_key.size(), epoch, f);
    return ref;
  }

  // one asynchronous search: lives from SearchAsync to its completion callback
  struct AsyncSearch {
    VectorGpu *self = nullptr;
    SearchDone done;
    std::unique_ptr<uint64_t[]> out;   // the library's answer: labels [count], distances [count]
    float *dist = nullptr;
    uint64_t n = 0;
    VkFilterRef filter;
    volatile int cancel_word = 0;
    bool watched = false;
    VkTokenWatch::Handle watch;
    static void Completed(void *user, int status) {     // a library thread; outputs are written
      std::unique_ptr<AsyncSearch> rq(static_cast<AsyncSearch *>(user));
      if (rq->watched) VkTokenWatch::Instance().Unregister(rq->watch);
      if (status != VK_OK) rq->done(VkCompletionStatus(status));
      else rq->done(rq->self->Reply(rq->dist, rq->out.get(), rq->n));
    }
  };
  // vk_batch_done_fn: a completer thread of the library, the outputs of every member are written
  static void BatchCompleted(void *self_p, const vk_completion *items, uint64_t n) {
    VectorGpu *self = static_cast<VectorGpu *>(self_p);
    std::vector<CompletedSearch> span;
    span.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
      std::unique_ptr<AsyncSearch> rq(static_cast<AsyncSearch *>(items[i].user));
      if (rq->watched) VkTokenWatch::Instance().Unregister(rq->watch);
      if (items[i].status != VK_OK) span.push_back(CompletedSearch{std::move(rq->done), VkCompletionStatus(items[i].status)});
      else span.push_back(CompletedSearch{std::move(rq->done), self->Reply(rq->dist, rq->out.get(), rq->n)});
    }
    if (self->bulk_done_) {
      self->bulk_done_(std::move(span));
    } else {
      for (CompletedSearch &c : span) c.done(std::move(c.result));
    }
  }
  BulkDone bulk_done_;

  vk_algo algo_;
  vk_index *ix_ = nullptr;
  bool owns_ = true;
  vk_index_params params_{};
  std::atomic<uint64_t> max_label_{0};
  std::atomic<uint64_t> filter_epoch_{0};
  mutable std::shared_mutex tracked_mu_;
  std::unordered_map<uint64_t, InternedStringPtr> tracked_;
};

// ---- the two classes IndexSchema instantiates (index_schema.cc: VectorFlat<float>::Create / VectorHNSW<float>::Create) ----
template <typename T>
class VectorGpuFlat final : public VectorGpu<T> {
 public:
  // VectorFlat<T>::Create, vector_flat.cc:53-74
  static absl::StatusOr<std::shared_ptr<VectorGpuFlat<T>>> Create(const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                  data_model::AttributeDataType attribute_data_type,
                                                                  uint32_t reader_threads = 256) {
    std::shared_ptr<VectorGpuFlat<T>> ix(new VectorGpuFlat<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type));
    if (auto st = ix->Open(FlatParams(proto), reader_threads); !st.ok()) return st;
    return ix;
  }
  // VectorFlat<T>::LoadFromRDB, vector_flat.cc:100-124 (called from index_schema.cc:200)
  static absl::StatusOr<std::shared_ptr<VectorGpuFlat<T>>> LoadFromRDB(ValkeyModuleCtx *ctx, const AttributeDataType *attribute_data_type,
                                                                       const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                       SupplementalContentChunkIter &&iter, uint32_t reader_threads = 256) {
    (void)ctx;
    std::shared_ptr<VectorGpuFlat<T>> ix(new VectorGpuFlat<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type->ToProto()));
    if (auto st = ix->OpenFromRDB(FlatParams(proto), reader_threads, std::move(iter)); !st.ok())
      return absl::InternalError(std::string("Error while loading a FLAT index: ") + std::string(st.message()));
    return ix;
  }
  static absl::StatusOr<std::shared_ptr<VectorGpuFlat<T>>> FromHandle(vk_index *handle, const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                      data_model::AttributeDataType attribute_data_type, uint32_t reader_threads = 256) {
    std::shared_ptr<VectorGpuFlat<T>> ix(new VectorGpuFlat<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type));
    if (auto st = ix->Adopt(handle, FlatParams(proto), reader_threads); !st.ok()) return st;
    return ix;
  }
  int GetBlockSize() const { return (int)this->params_.block_size; }

 protected:
  void ToProtoImpl(data_model::VectorIndex *proto) const override { proto->mutable_flat_algorithm()->set_block_size(this->params_.block_size); }
  int RespondWithInfoImpl(ValkeyModuleCtx *ctx) const override {   // vector_flat.cc:100-124 + the device counters
    ValkeyModule_ReplyWithSimpleString(ctx, "data_type");
    ValkeyModule_ReplyWithSimpleString(ctx, "FLOAT32");
    ValkeyModule_ReplyWithSimpleString(ctx, "algorithm");
    ValkeyModule_ReplyWithSimpleString(ctx, "FLAT");
    ValkeyModule_ReplyWithSimpleString(ctx, "block_size");
    ValkeyModule_ReplyWithLongLong(ctx, this->params_.block_size);
    return 6 + this->RespondWithCounters(ctx);
  }

 private:
  static vk_index_params FlatParams(const data_model::VectorIndex &proto) {
    vk_index_params p{};
    p.initial_cap = proto.initial_cap();
    p.block_size = proto.flat_algorithm().block_size();
    return p;
  }
  VectorGpuFlat(int dimensions, data_model::DistanceMetric metric, absl::string_view attribute_identifier, data_model::AttributeDataType adt)
      : VectorGpu<T>(IndexerType::kFlat, VK_ALGO_FLAT, dimensions, metric, attribute_identifier, adt) {}
};

template <typename T>
class VectorGpuHNSW final : public VectorGpu<T> {
 public:
  // VectorHNSW<T>::Create, vector_hnsw.cc:84-108
  static absl::StatusOr<std::shared_ptr<VectorGpuHNSW<T>>> Create(const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                  data_model::AttributeDataType attribute_data_type,
                                                                  bool allow_replace_deleted = false, uint32_t block_size = 10240,
                                                                  uint32_t reader_threads = 256) {
    std::shared_ptr<VectorGpuHNSW<T>> ix(new VectorGpuHNSW<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type));
    if (auto st = ix->Open(HnswParams(proto, allow_replace_deleted, block_size, true), reader_threads); !st.ok()) return st;
    return ix;
  }
  // VectorHNSW<T>::LoadFromRDB, vector_hnsw.cc:135-166 (called from index_schema.cc:175): initial_cap keeps the definition's
  // capacity for an empty stream, M is checked against the stream, `validate` is hnsw-validation-enable, ef_runtime is not
  // persisted and comes from the definition
  static absl::StatusOr<std::shared_ptr<VectorGpuHNSW<T>>> LoadFromRDB(ValkeyModuleCtx *ctx, const AttributeDataType *attribute_data_type,
                                                                       const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                       SupplementalContentChunkIter &&iter, bool allow_replace_deleted = false,
                                                                       bool validate = true, uint32_t block_size = 10240,
                                                                       uint32_t reader_threads = 256) {
    (void)ctx;
    std::shared_ptr<VectorGpuHNSW<T>> ix(new VectorGpuHNSW<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type->ToProto()));
    if (auto st = ix->OpenFromRDB(HnswParams(proto, allow_replace_deleted, block_size, validate), reader_threads, std::move(iter)); !st.ok())
      return absl::InternalError(std::string("HNSWLib error while loading an index: ") + std::string(st.message()));
    return ix;
  }
  static absl::StatusOr<std::shared_ptr<VectorGpuHNSW<T>>> FromHandle(vk_index *handle, const data_model::VectorIndex &proto, absl::string_view attribute_identifier,
                                                                      data_model::AttributeDataType attribute_data_type, uint32_t reader_threads = 256) {
    std::shared_ptr<VectorGpuHNSW<T>> ix(new VectorGpuHNSW<T>((int)proto.dimension_count(), proto.distance_metric(), attribute_identifier,
                                                              attribute_data_type));
    if (auto st = ix->Adopt(handle, HnswParams(proto, false, 10240, true), reader_threads); !st.ok()) return st;
    return ix;
  }
  int GetM() const { return (int)this->params_.m; }
  int GetEfConstruction() const { return (int)this->params_.ef_construction; }
  size_t GetEfRuntime() const { return this->params_.ef_runtime; }

 protected:
  void ToProtoImpl(data_model::VectorIndex *proto) const override {
    auto *h = proto->mutable_hnsw_algorithm();
    h->set_m(this->params_.m);
    h->set_ef_construction(this->params_.ef_construction);
    h->set_ef_runtime(this->params_.ef_runtime);
  }
  int RespondWithInfoImpl(ValkeyModuleCtx *ctx) const override {   // vector_hnsw.cc:202-229 + the device counters
    ValkeyModule_ReplyWithSimpleString(ctx, "data_type");
    ValkeyModule_ReplyWithSimpleString(ctx, "FLOAT32");
    ValkeyModule_ReplyWithSimpleString(ctx, "algorithm");
    ValkeyModule_ReplyWithSimpleString(ctx, "HNSW");
    ValkeyModule_ReplyWithSimpleString(ctx, "m");
    ValkeyModule_ReplyWithLongLong(ctx, GetM());
    ValkeyModule_ReplyWithSimpleString(ctx, "ef_construction");
    ValkeyModule_ReplyWithLongLong(ctx, GetEfConstruction());
    ValkeyModule_ReplyWithSimpleString(ctx, "ef_runtime");
    ValkeyModule_ReplyWithLongLong(ctx, (long long)GetEfRuntime());
    return 10 + this->RespondWithCounters(ctx);
  }

 private:
  static vk_index_params HnswParams(const data_model::VectorIndex &proto, bool allow_replace_deleted, uint32_t block_size, bool validate) {
    vk_index_params p{};
    p.initial_cap = proto.initial_cap();
    p.block_size = block_size;                                  // options::GetHNSWBlockSize()
    p.m = proto.hnsw_algorithm().m();
    p.ef_construction = proto.hnsw_algorithm().ef_construction();
    p.ef_runtime = proto.hnsw_algorithm().ef_runtime();
    p.allow_replace_deleted = allow_replace_deleted ? 1u : 0u;  // options::GetHNSWAllowReplaceDeleted()
    p.load_skip_validation = validate ? 0u : 1u;                // options::GetHNSWValidationEnable()
    return p;
  }
  VectorGpuHNSW(int dimensions, data_model::DistanceMetric metric, absl::string_view attribute_identifier, data_model::AttributeDataType adt)
      : VectorGpu<T>(IndexerType::kHNSW, VK_ALGO_HNSW, dimensions, metric, attribute_identifier, adt) {}
};

}  // namespace valkey_search::indexes
#endif  // VK_VECTOR_ADAPTOR_H_
