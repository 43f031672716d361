// vector_index.cc -- see vector_index.h.  Line references: src/indexes/vector_base.cc unless noted.
#include "vector_index.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace vsa {

namespace cancel {
namespace {
struct Manual : Base {
  std::atomic<int> flag{0};
  bool IsCancelled() override { return flag.load() != 0; }
  void Cancel() override { flag.store(1); }
  const volatile int *Flag() const override { return reinterpret_cast<const volatile int *>(&flag); }
};
}  // namespace
Token Make() { return std::make_shared<Manual>(); }
}  // namespace cancel

namespace indexes {

namespace {
constexpr float kDefaultMagnitude = -1.0f;

Status FromVk(int rc) {
  std::string msg = vk_last_error();
  switch (rc) {
    case VK_OK: return OkStatus();
    case VK_ERR_INVALID: return InvalidArgumentError(msg);
    case VK_ERR_NOT_FOUND: return NotFoundError(msg);
    case VK_ERR_CANCELLED: return CancelledError(msg);
    default: return InternalError(msg);
  }
}

// :112-124 CopyAndNormalizeEmbedding, sequential f32
float CopyAndNormalizeEmbedding(float *dst, const float *src, size_t size) {
  float magnitude = 0.0f;
  for (size_t i = 0; i < size; i++) magnitude += src[i] * src[i];
  magnitude = sqrtf(magnitude);
  const float norm = (magnitude == 0.0f) ? 1.0f : (1.0f / magnitude);
  for (size_t i = 0; i < size; i++) dst[i] = norm * src[i];
  return magnitude;
}
}  // namespace

std::vector<char> NormalizeEmbedding(std::string_view record, size_t type_size, float *magnitude) {
  std::vector<char> ret(record.size());
  (void)type_size;
  std::vector<float> tmp(record.size() / sizeof(float));
  memcpy(tmp.data(), record.data(), tmp.size() * sizeof(float));
  float m = CopyAndNormalizeEmbedding(reinterpret_cast<float *>(ret.data()), tmp.data(), tmp.size());
  if (magnitude) *magnitude = m;
  return ret;
}

VectorBase::VectorBase(IndexerType type, int dimensions, DistanceMetric metric, vk_index *ix)
    : type_(type), dimensions_(dimensions), metric_(metric), normalize_(metric == DistanceMetric::kCosine), ix_(ix) {}   // :140-150

VectorBase::~VectorBase() { vk_index_destroy(ix_); }

bool VectorBase::IsTracked(const std::string &key) const {
  std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
  return tracked_metadata_by_key_.count(key) != 0;
}
size_t VectorBase::GetTrackedKeyCount() const {
  std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
  return tracked_metadata_by_key_.size();
}
size_t VectorBase::GetCapacity() const {
  vk_index_stats s;
  if (vk_index_get_stats(ix_, &s) != VK_OK) return 0;
  return s.capacity;
}

// AddRecordImpl with the "exceeds the specified limit" -> ResizeIfFull -> retry loop
// (vector_flat.cc:157-176, vector_hnsw.cc:176-197)
Status VectorBase::AddRecordImpl(uint64_t internal_id, std::string_view record) {
  for (;;) {
    int rc = vk_index_add(ix_, internal_id, record.data());
    if (rc == VK_OK) return OkStatus();
    if (rc == VK_ERR_CAPACITY) {
      Status s = ResizeIfFull();
      if (!s.ok()) return s;
      continue;
    }
    return InternalError(std::string("Error while adding a record: ") + vk_last_error());
  }
}

StatusOr<RecordResult> VectorBase::AddRecord(const std::string &key, std::string_view record) {   // :168-187
  if (!IsValidSizeVector(record)) return RecordResult::kInvalidData;
  float magnitude = kDefaultMagnitude;
  std::vector<char> norm;
  if (normalize_) {
    norm = NormalizeEmbedding(record, sizeof(float), &magnitude);
    record = std::string_view(norm.data(), norm.size());
  }
  uint64_t id;
  {   // TrackKey :340-358
    if (key.empty()) return InvalidArgumentError("key can't be empty");
    std::unique_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
    id = inc_id_++;
    auto ins = tracked_metadata_by_key_.insert({key, {id, magnitude}});
    if (!ins.second) return InvalidArgumentError("Embedding id already exists: " + key);
    key_by_internal_id_.insert({id, key});
  }
  Status add = AddRecordImpl(id, record);
  if (!add.ok()) {
    std::unique_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
    tracked_metadata_by_key_.erase(key);
    key_by_internal_id_.erase(id);
    return add;
  }
  return RecordResult::kAdded;
}

StatusOr<RecordResult> VectorBase::ModifyRecord(const std::string &key, std::string_view record) {   // :225-256
  if (!IsValidSizeVector(record)) {
    (void)RemoveRecord(key, DeletionType::kRecord);
    return RecordResult::kInvalidData;
  }
  float magnitude = kDefaultMagnitude;
  std::vector<char> norm;
  if (normalize_) {
    norm = NormalizeEmbedding(record, sizeof(float), &magnitude);
    record = std::string_view(norm.data(), norm.size());
  }
  uint64_t id;
  {
    if (key.empty()) return InvalidArgumentError("key can't be empty");
    std::unique_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
    auto it = tracked_metadata_by_key_.find(key);
    if (it == tracked_metadata_by_key_.end()) return InvalidArgumentError("Embedding id not found: " + key);
    it->second.magnitude = magnitude;
    id = it->second.internal_id;
  }
  // IsVectorMatch (:374-382): identical bytes -> nothing to re-index
  std::vector<float> cur(dimensions_);
  if (vk_index_get_row(ix_, id, cur.data()) == VK_OK && memcmp(cur.data(), record.data(), record.size()) == 0)
    return RecordResult::kMissing;
  int rc = vk_index_add(ix_, id, record.data());     // same label = in-place update
  if (rc != VK_OK) return InternalError(std::string("Error while modifying a record: ") + vk_last_error());
  return RecordResult::kAdded;
}

StatusOr<bool> VectorBase::RemoveRecord(const std::string &key, DeletionType) {   // :299-331
  uint64_t id;
  {
    if (key.empty()) return false;
    std::unique_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
    auto it = tracked_metadata_by_key_.find(key);
    if (it == tracked_metadata_by_key_.end()) return false;
    id = it->second.internal_id;
    tracked_metadata_by_key_.erase(it);
    key_by_internal_id_.erase(id);
  }
  int rc = vk_index_remove(ix_, id);
  if (rc != VK_OK) return InternalError(std::string("Error while removing a record: ") + vk_last_error());
  return true;
}

std::vector<Neighbor> VectorBase::CreateReply(const float *dist, const uint64_t *labels, uint64_t n) const {   // :258-277
  std::vector<Neighbor> ret;
  ret.reserve(n);
  for (uint64_t i = 0; i < n; ++i) {
    auto it = key_by_internal_id_.find(labels[i]);
    if (it == key_by_internal_id_.end()) continue;   // unknown label: dropped
    ret.push_back(Neighbor{it->second, dist[i]});
  }
  return ret;   // the ABI already returns ascending (distance,label): pop + reverse of the reference heap
}

StatusOr<std::vector<Neighbor>> VectorBase::SearchImpl(std::string_view query, uint64_t count, cancel::Token &token,
                                                       const KeyPredicate *filter, std::optional<size_t> ef_runtime,
                                                       bool enable_partial_results) const {
  if (!IsValidSizeVector(query)) return InvalidArgumentError("query vector has the wrong size");
  std::vector<char> norm;
  if (normalize_) {   // vector_flat.cc:244-250, vector_hnsw.cc:337-342
    norm = NormalizeEmbedding(query, sizeof(float));
    query = std::string_view(norm.data(), norm.size());
  }
  std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
  std::vector<uint64_t> bits;
  uint64_t nbits = 0;
  if (filter) {   // InlineVectorFilter (search.cc:103-134) evaluated once per tracked key
    nbits = inc_id_;
    bits.assign((nbits + 63) / 64, 0);
    for (const auto &kv : tracked_metadata_by_key_)
      if ((*filter)(kv.first)) bits[kv.second.internal_id >> 6] |= 1ull << (kv.second.internal_id & 63);
  }
  if (count == 0) return std::vector<Neighbor>{};
  std::vector<float> dist(count);
  std::vector<uint64_t> labels(count);
  uint64_t n = 0;
  int rc = vk_index_search(ix_, query.data(), count, ef_runtime.value_or(0), filter ? bits.data() : nullptr, nbits,
                           token ? token->Flag() : nullptr, enable_partial_results || type_ == IndexerType::kFlat,
                           dist.data(), labels.data(), &n);
  if (rc == VK_ERR_CANCELLED) return CancelledError("Search operation cancelled due to timeout");
  if (rc != VK_OK) return InternalError(vk_last_error());
  return CreateReply(dist.data(), labels.data(), n);
}

StatusOr<std::pair<float, uint64_t>> VectorBase::ComputeDistanceFromRecord(const std::string &key, std::string_view query) const {
  uint64_t id;
  {
    std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
    auto it = tracked_metadata_by_key_.find(key);
    if (it == tracked_metadata_by_key_.end()) return InvalidArgumentError("Record was not found");
    id = it->second.internal_id;
  }
  float d;
  int rc = vk_index_distance(ix_, id, query.data(), &d);
  if (rc != VK_OK) return InternalError("Couldn't find internal id: " + std::to_string(id));
  return std::make_pair(d, id);
}

StatusOr<std::vector<Neighbor>> VectorBase::SearchPrefiltered(std::string_view query, uint64_t count,
                                                              const std::vector<std::string> &keys) const {
  std::vector<char> norm;
  if (normalize_) {   // search.cc:463-468
    norm = NormalizeEmbedding(query, sizeof(float));
    query = std::string_view(norm.data(), norm.size());
  }
  std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
  std::vector<uint64_t> ids;
  ids.reserve(keys.size());
  for (const auto &k : keys) {
    auto it = tracked_metadata_by_key_.find(k);
    if (it != tracked_metadata_by_key_.end()) ids.push_back(it->second.internal_id);
  }
  std::vector<float> dist(count ? count : 1);
  std::vector<uint64_t> labels(count ? count : 1);
  uint64_t n = 0;
  int rc = vk_index_search_labels(ix_, query.data(), count, ids.data(), ids.size(), dist.data(), labels.data(), &n);
  if (rc != VK_OK) return FromVk(rc);
  return CreateReply(dist.data(), labels.data(), n);
}

StatusOr<std::vector<char>> VectorBase::GetValue(const std::string &key) const {   // :279-297
  std::shared_lock<std::shared_mutex> lk(key_to_metadata_mutex_);
  auto it = tracked_metadata_by_key_.find(key);
  if (it == tracked_metadata_by_key_.end()) return NotFoundError("Record was not found");
  std::vector<char> result((size_t)dimensions_ * sizeof(float));
  if (vk_index_get_row(ix_, it->second.internal_id, result.data()) != VK_OK) return InternalError(vk_last_error());
  if (normalize_) {
    if (it->second.magnitude < 0) return InternalError("Magnitude is not initialized");
    float *f = reinterpret_cast<float *>(result.data());   // DenormalizeVector (vector_externalizer.cc:30-39)
    for (int i = 0; i < dimensions_; ++i) f[i] *= it->second.magnitude;
  }
  return result;
}

Status VectorBase::SaveIndex(vk_write_chunk_fn fn, void *user) const { return FromVk(vk_index_save(ix_, fn, user)); }

// ---- VectorFlat ------------------------------------------------------------------------------------
static vk_metric ToVk(DistanceMetric m) {
  return m == DistanceMetric::kL2 ? VK_METRIC_L2 : m == DistanceMetric::kIP ? VK_METRIC_IP : VK_METRIC_COSINE;
}

template <typename T>
StatusOr<std::shared_ptr<VectorFlat<T>>> VectorFlat<T>::Create(const FlatParams &p) {
  vk_index_params vp{};
  vp.struct_size = sizeof(vp);
  vp.algo = VK_ALGO_FLAT;
  vp.metric = ToVk(p.metric);
  vp.dtype = VK_DTYPE_F32;
  vp.dim = (uint32_t)p.dimensions;
  vp.block_size = p.block_size;
  vp.initial_cap = p.initial_cap;
  vp.device_id = -1;
  vk_index *ix = nullptr;
  int rc = vk_index_create(&vp, &ix);
  if (rc != VK_OK) return InternalError(std::string("Error while creating a FLAT index: ") + vk_last_error());
  return std::shared_ptr<VectorFlat<T>>(new VectorFlat<T>(p, ix));
}

template <typename T>
Status VectorFlat<T>::ResizeIfFull() {   // vector_flat.cc:137-155
  if (block_size_ == 0) return InternalError("Cannot resize FLAT index: block_size is 0");
  return FromVk(vk_index_resize(ix_, GetCapacity() + block_size_));
}

template <typename T>
StatusOr<std::vector<Neighbor>> VectorFlat<T>::Search(std::string_view query, uint64_t count, cancel::Token &token,
                                                      const KeyPredicate *filter) const {
  return SearchImpl(query, count, token, filter, std::nullopt, true);
}

// ---- VectorHNSW ------------------------------------------------------------------------------------
template <typename T>
StatusOr<std::shared_ptr<VectorHNSW<T>>> VectorHNSW<T>::Create(const HnswParams &p) {
  vk_index_params vp{};
  vp.struct_size = sizeof(vp);
  vp.algo = VK_ALGO_HNSW;
  vp.metric = ToVk(p.metric);
  vp.dtype = VK_DTYPE_F32;
  vp.dim = (uint32_t)p.dimensions;
  vp.block_size = p.hnsw_block_size;
  vp.initial_cap = p.initial_cap;
  vp.m = p.m;
  vp.ef_construction = p.ef_construction;
  vp.ef_runtime = p.ef_runtime;
  vp.allow_replace_deleted = p.allow_replace_deleted;
  vp.random_seed = 100;
  vp.device_id = -1;
  vk_index *ix = nullptr;
  int rc = vk_index_create(&vp, &ix);
  if (rc != VK_OK) return InternalError(std::string("HNSWLib error while creating a record: ") + vk_last_error());
  return std::shared_ptr<VectorHNSW<T>>(new VectorHNSW<T>(p, ix));
}

template <typename T>
Status VectorHNSW<T>::ResizeIfFull() {   // vector_hnsw.cc:238-271
  return FromVk(vk_index_resize(ix_, GetCapacity() + block_size_));
}

template <typename T>
StatusOr<std::vector<Neighbor>> VectorHNSW<T>::Search(std::string_view query, uint64_t count, cancel::Token &token,
                                                      const KeyPredicate *filter, std::optional<size_t> ef_runtime,
                                                      bool enable_partial_results) const {
  return SearchImpl(query, count, token, filter, ef_runtime, enable_partial_results);
}

template class VectorFlat<float>;
template class VectorHNSW<float>;

}  // namespace indexes

namespace query {
bool UsePreFiltering(size_t estimated_num_of_keys, const indexes::VectorBase *vector_index) {   // planner.cc:21-45
  if (vector_index->GetIndexerType() == indexes::IndexerType::kFlat) return true;
  const size_t N = vector_index->GetTrackedKeyCount();
  return estimated_num_of_keys <= kPrefilteringThresholdRatio * N;
}
}  // namespace query

}  // namespace vsa
