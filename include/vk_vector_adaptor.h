// vk_vector_adaptor.h -- the reference-side binding in the shape SURVEY section 8(b) asks for: classes DERIVED FROM
// valkey_search::indexes::VectorBase with the overrides of VectorFlat<T> / VectorHNSW<T>
// (src/indexes/vector_flat.h:37-63, vector_hnsw.h:36-73; the pure virtuals of vector_base.h:129-282), implemented on the
// C ABI of vk_index.h.  Source-only: in the module it is compiled with VK_ADAPTOR_IN_TREE defined (real headers); this
// repository compiles it against tests/helpers/mock_valkey_search.h with -Wall -Werror (tests/test_abi_symbols.py) and
// drives it on a GPU (tests/test_facade_gpu.py), so a drift between these signatures and the interface is a build error.
//
// What differs from the hnswlib-backed classes, and why:
//   * Search() takes the same arguments.  The filter functor is evaluated ONCE per call over the label space into an
//     allow-bitmap (a functor cannot be called from the device; planner.cc:21-45 sends small filtered sets down the
//     pre-filter path instead, which never reaches Search).  The cancellation token is polled by the CALLING thread
//     while it waits for the completion of vk_index_search_submit and relayed as the ABI's cancel word; the kernels
//     stop within about a millisecond (vk_index.h).
//   * the search is SUBMITTED, not called: the reader-pool thread parks on a condition variable while the library
//     coalesces the pool's concurrent FT.SEARCHes into device batches and keeps two of them in flight
//     (src/query/search.cc:886-910 queues one request per reader thread; here the queue is the library's).
//   * GetValueImpl serves the tracked (interned) vector: the rows themselves live in HBM.
//   * RespondWithInfoImpl appends the counters metrics.h:40-50,75-80 and hnswalg.h:98-99,1199 expose
//     (vk_index_stats: searches, errors by kind, distance computations, hops, reclaimable bytes, latency histogram).
#ifndef VK_VECTOR_ADAPTOR_H_
#define VK_VECTOR_ADAPTOR_H_

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#ifdef VK_ADAPTOR_IN_TREE
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

template <typename T>
class VectorGpu : public VectorBase {
 public:
  ~VectorGpu() override { vk_index_destroy(ix_); }
  size_t GetDataTypeSize() const override { return sizeof(T); }
  size_t GetCapacity() const override { return Stats().capacity; }
  uint64_t GetMaxInternalLabel() const override { return max_label_.load(std::memory_order_relaxed); }
  size_t GetLabelCount() const override { return Stats().count; }
  int GetDimensions() const { return dimensions_; }
  vk_index *handle() const { return ix_; }
  const vk_index_params &params() const { return params_; }   // LoadFromRDB passes them to vk_index_load

  // VectorFlat<T>::Search (vector_flat.cc:224-254) / VectorHNSW<T>::Search (vector_hnsw.cc:313-347)
  absl::StatusOr<std::vector<Neighbor>> Search(absl::string_view query, uint64_t count, cancel::Token &cancellation_token,
                                               std::unique_ptr<hnswlib::BaseFilterFunctor> filter = nullptr,
                                               std::optional<size_t> ef_runtime = std::nullopt, bool enable_partial_results = false) {
    if (query.size() != (size_t)GetVectorDataSize()) return absl::InvalidArgumentError("query vector of the wrong size");
    std::vector<T> normalised;
    const void *q = query.data();
    if (normalize_) {   // CopyAndNormalizeEmbedding (vector_base.cc:112-124): sequential f32, 1 / magnitude, zero vector unchanged
      const T *src = reinterpret_cast<const T *>(query.data());
      normalised.resize((size_t)dimensions_);
      T magnitude = 0;
      for (int i = 0; i < dimensions_; ++i) magnitude += src[i] * src[i];
      magnitude = std::sqrt(magnitude);
      const T norm = magnitude == (T)0 ? (T)1 : (T)1 / magnitude;
      for (int i = 0; i < dimensions_; ++i) normalised[(size_t)i] = norm * src[i];
      q = normalised.data();
    }
    std::vector<uint64_t> allow;
    uint64_t allow_nbits = 0;
    if (filter) {                                              // the materialised BaseFilterFunctor (hnswlib.h:144-149)
      allow_nbits = GetMaxInternalLabel() + 1;
      allow.assign((allow_nbits + 63) / 64, 0);
      for (uint64_t id = 0; id < allow_nbits; ++id)
        if ((*filter)(id)) allow[id >> 6] |= 1ull << (id & 63);
    }
    const uint64_t k = count;                                  // (FLAT clamps to the element count inside, vector_flat.cc:234-236)
    std::vector<float> dist(k ? k : 1);
    std::vector<uint64_t> label(k ? k : 1);
    uint64_t n = 0;
    Waiter w;
    // (a token that is already up travels with the request; after that the parked caller polls it)
    volatile int cancel_word = cancellation_token && cancellation_token->IsCancelled() ? 1 : 0;
    const int rc = vk_index_search_submit(ix_, q, k, ef_runtime.value_or(0), filter ? allow.data() : nullptr, allow_nbits, &cancel_word,
                                          enable_partial_results ? 1 : 0, dist.data(), label.data(), &n, &Waiter::Done, &w);
    if (rc != VK_OK) return VkToStatus(rc);
    {
      std::unique_lock<std::mutex> lk(w.mu);
      while (!w.done) {                                        // the token is time-based (cancel.h): poll it while parked
        w.cv.wait_for(lk, std::chrono::microseconds(250));
        if (!w.done && cancellation_token && cancellation_token->IsCancelled()) cancel_word = 1;
      }
    }
    // (HNSW without partial results answers a raised token with VK_ERR_CANCELLED -> CancelledError, vector_hnsw.cc:327-329;
    //  FLAT returns what its scan had, like bruteforce.h:125-129 -- VectorFlat::Search has no such branch)
    if (w.status != VK_OK) return VkToStatus(w.status);
    std::vector<Neighbor> out;                                 // CreateReply, vector_base.cc:258-277: ascending by distance
    out.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
      auto key = GetKeyDuringSearch(label[i]);
      if (!key.ok()) continue;                                 // (a key removed since: skipped like CreateReply does)
      out.emplace_back(key.value(), dist[i]);
    }
    return out;
  }

 protected:
  VectorGpu(IndexerType type, vk_algo algo, int dimensions, data_model::DistanceMetric metric, absl::string_view attribute_identifier,
            data_model::AttributeDataType attribute_data_type)
      : VectorBase(type, dimensions, attribute_data_type, attribute_identifier), algo_(algo) {
    distance_metric_ = metric;
    normalize_ = metric == data_model::DISTANCE_METRIC_COSINE;  // VectorBase::Init, vector_base.cc:140-150
  }
  absl::Status Open(vk_index_params p, uint32_t reader_threads) {
    p.struct_size = sizeof p;
    p.algo = algo_;
    p.dtype = VK_DTYPE_F32;
    p.dim = (uint32_t)dimensions_;
    p.metric = distance_metric_ == data_model::DISTANCE_METRIC_L2   ? VK_METRIC_L2
               : distance_metric_ == data_model::DISTANCE_METRIC_IP ? VK_METRIC_IP
                                                                    : VK_METRIC_COSINE;
    p.random_seed = 100;
    p.device_id = -1;
    params_ = p;
    if (int rc = vk_index_create(&p, &ix_); rc != VK_OK) return VkToStatus(rc);
    // the reader pool's single-query calls become device batches; the submit path needs it on
    return VkToStatus(vk_index_set_coalescing(ix_, reader_threads > 1 ? reader_threads : 256, 200));
  }

  // AddRecordImpl with the resize-and-retry loop of vector_flat.cc:137-176 / vector_hnsw.cc:238-271
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
        {"gpu_queue_depth", s.queued_now},          {"gpu_rejected_busy", s.rejected}};
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

  struct Waiter {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int status = 0;
    static void Done(void *user, int status) {
      auto *w = static_cast<Waiter *>(user);
      std::lock_guard<std::mutex> l(w->mu);
      w->status = status;
      w->done = true;
      w->cv.notify_one();
    }
  };

  vk_algo algo_;
  vk_index *ix_ = nullptr;
  vk_index_params params_{};
  std::atomic<uint64_t> max_label_{0};
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
    vk_index_params p{};
    p.initial_cap = proto.initial_cap();
    p.block_size = proto.flat_algorithm().block_size();
    if (auto st = ix->Open(p, reader_threads); !st.ok()) return st;
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
    vk_index_params p{};
    p.initial_cap = proto.initial_cap();
    p.block_size = block_size;                                  // options::GetHNSWBlockSize()
    p.m = proto.hnsw_algorithm().m();
    p.ef_construction = proto.hnsw_algorithm().ef_construction();
    p.ef_runtime = proto.hnsw_algorithm().ef_runtime();
    p.allow_replace_deleted = allow_replace_deleted ? 1u : 0u;  // options::GetHNSWAllowReplaceDeleted()
    if (auto st = ix->Open(p, reader_threads); !st.ok()) return st;
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
  VectorGpuHNSW(int dimensions, data_model::DistanceMetric metric, absl::string_view attribute_identifier, data_model::AttributeDataType adt)
      : VectorGpu<T>(IndexerType::kHNSW, VK_ALGO_HNSW, dimensions, metric, attribute_identifier, adt) {}
};

}  // namespace valkey_search::indexes
#endif  // VK_VECTOR_ADAPTOR_H_
