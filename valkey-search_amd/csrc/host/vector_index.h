// vector_index.h -- host-side C++ mirror of valkey-search's vector index classes, implemented on the
// C ABI of include/vk_index.h (the reference's toolchain deps -- abseil, protobuf, the module SDK --
// are absent in this image, so the host side above the ABI is stand-alone C++ with the same names,
// argument meaning and result codes).
//
//   vsa::indexes::VectorBase          <- src/indexes/vector_base.{h,cc}
//   vsa::indexes::VectorFlat<float>   <- src/indexes/vector_flat.{h,cc}
//   vsa::indexes::VectorHNSW<float>   <- src/indexes/vector_hnsw.{h,cc}
//   vsa::query::UsePreFiltering       <- src/query/planner.cc:21-45
//   vsa::query::CalcBestMatchingPrefilteredKeys <- src/query/search.cc:457-481
// Keys are std::string (InternedStringPtr in the reference); absl::Status is vsa::Status with the
// same codes; filters reach the device as a bitmap over internal ids, built here from a predicate
// over keys (what InlineVectorFilter evaluates per candidate, src/query/search.cc:103-134).
#pragma once
#include <stdint.h>

#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <string>
#include <string_view>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/vk_index.h"

namespace vsa {

enum class StatusCode { kOk = 0, kCancelled = 1, kInvalidArgument = 3, kNotFound = 5, kAlreadyExists = 6, kInternal = 13 };
class Status {
 public:
  Status() = default;
  Status(StatusCode c, std::string m) : code_(c), msg_(std::move(m)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string &message() const { return msg_; }
 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(std::string m) { return Status(StatusCode::kInvalidArgument, std::move(m)); }
inline Status InternalError(std::string m) { return Status(StatusCode::kInternal, std::move(m)); }
inline Status NotFoundError(std::string m) { return Status(StatusCode::kNotFound, std::move(m)); }
inline Status CancelledError(std::string m) { return Status(StatusCode::kCancelled, std::move(m)); }

template <class T>
class StatusOr {
 public:
  StatusOr(Status s) : status_(std::move(s)) {}           // NOLINT: implicit like absl
  StatusOr(T v) : value_(std::move(v)) {}                 // NOLINT
  bool ok() const { return status_.ok(); }
  const Status &status() const { return status_; }
  T &value() { return *value_; }
  const T &value() const { return *value_; }
  T &operator*() { return *value_; }
  T *operator->() { return &*value_; }
 private:
  Status status_;
  std::optional<T> value_;
};

namespace cancel {           // src/utils/cancel.h
struct Base {
  virtual ~Base() = default;
  virtual bool IsCancelled() = 0;
  virtual void Cancel() = 0;
  // host word handed to the device side (vk_index_search cancel_flag)
  virtual const volatile int *Flag() const = 0;
};
using Token = std::shared_ptr<Base>;
Token Make();                // never cancels until Cancel()
}  // namespace cancel

namespace indexes {

enum class IndexerType { kHNSW, kFlat };
enum class DeletionType { kRecord, kIdentifier, kNone };
enum class RecordResult { kAdded, kMissing, kInvalidData };   // index_base.h:46-56
enum class DistanceMetric { kL2, kIP, kCosine };

struct Neighbor {            // vector_base.h:58-
  std::string external_id;
  float distance = 0.f;
};

// The materialised BaseFilterFunctor: predicate over keys -> bitmap over internal ids.
using KeyPredicate = std::function<bool(const std::string &key)>;

std::vector<char> NormalizeEmbedding(std::string_view record, size_t type_size, float *magnitude = nullptr);  // vector_base.cc:126-138

class VectorBase {
 public:
  virtual ~VectorBase();
  StatusOr<RecordResult> AddRecord(const std::string &key, std::string_view record);          // vector_base.cc:168-187
  StatusOr<bool> RemoveRecord(const std::string &key, DeletionType = DeletionType::kNone);     // :299-308
  StatusOr<RecordResult> ModifyRecord(const std::string &key, std::string_view record);        // :225-256
  bool IsTracked(const std::string &key) const;
  size_t GetTrackedKeyCount() const;
  IndexerType GetIndexerType() const { return type_; }
  bool GetNormalize() const { return normalize_; }
  int GetDimensions() const { return dimensions_; }
  size_t GetDataTypeSize() const { return sizeof(float); }
  size_t GetCapacity() const;
  StatusOr<std::vector<char>> GetValue(const std::string &key) const;                           // :279-297 (denormalised for COSINE)
  StatusOr<std::pair<float, uint64_t>> ComputeDistanceFromRecord(const std::string &key, std::string_view query) const;  // :502-507
  // exact kNN over a key list with the AddPrefilteredKey heap rule (:509-530), device distances
  StatusOr<std::vector<Neighbor>> SearchPrefiltered(std::string_view query, uint64_t count, const std::vector<std::string> &keys) const;
  uint64_t GetMaxInternalLabel() const { return inc_id_ ? inc_id_ - 1 : 0; }
  Status SaveIndex(vk_write_chunk_fn fn, void *user) const;

 protected:
  VectorBase(IndexerType type, int dimensions, DistanceMetric metric, vk_index *ix);
  StatusOr<std::vector<Neighbor>> SearchImpl(std::string_view query, uint64_t count, cancel::Token &token,
                                             const KeyPredicate *filter, std::optional<size_t> ef_runtime,
                                             bool enable_partial_results) const;
  std::vector<Neighbor> CreateReply(const float *dist, const uint64_t *labels, uint64_t n) const;  // :258-277
  bool IsValidSizeVector(std::string_view record) const {
    return record.size() % sizeof(float) == 0 && (int)(record.size() / sizeof(float)) == dimensions_;
  }
  virtual Status ResizeIfFull() = 0;
  Status AddRecordImpl(uint64_t internal_id, std::string_view record);

  IndexerType type_;
  int dimensions_;
  DistanceMetric metric_;
  bool normalize_ = false;
  vk_index *ix_ = nullptr;
  mutable std::shared_mutex key_to_metadata_mutex_;
  struct TrackedKeyMetadata { uint64_t internal_id; float magnitude; };
  std::unordered_map<std::string, TrackedKeyMetadata> tracked_metadata_by_key_;
  std::unordered_map<uint64_t, std::string> key_by_internal_id_;
  uint64_t inc_id_ = 0;
};

struct FlatParams { int dimensions; DistanceMetric metric; uint64_t initial_cap; uint32_t block_size = 1024; };
struct HnswParams { int dimensions; DistanceMetric metric; uint64_t initial_cap; uint32_t m = 16; uint32_t ef_construction = 200;
                    uint32_t ef_runtime = 10; uint32_t hnsw_block_size = 10240; bool allow_replace_deleted = false; };

template <typename T>
class VectorFlat : public VectorBase {
 public:
  static StatusOr<std::shared_ptr<VectorFlat<T>>> Create(const FlatParams &p);                 // vector_flat.cc:53-74
  int GetBlockSize() const { return block_size_; }
  StatusOr<std::vector<Neighbor>> Search(std::string_view query, uint64_t count, cancel::Token &cancellation_token,
                                         const KeyPredicate *filter = nullptr) const;          // :224-254
 protected:
  Status ResizeIfFull() override;                                                               // :137-155
 private:
  VectorFlat(const FlatParams &p, vk_index *ix) : VectorBase(IndexerType::kFlat, p.dimensions, p.metric, ix), block_size_(p.block_size) {}
  uint32_t block_size_;
};

template <typename T>
class VectorHNSW : public VectorBase {
 public:
  static StatusOr<std::shared_ptr<VectorHNSW<T>>> Create(const HnswParams &p);                 // vector_hnsw.cc:84-108
  int GetM() const { return m_; }
  int GetEfConstruction() const { return ef_construction_; }
  size_t GetEfRuntime() const { return ef_runtime_; }
  StatusOr<std::vector<Neighbor>> Search(std::string_view query, uint64_t count, cancel::Token &cancellation_token,
                                         const KeyPredicate *filter = nullptr, std::optional<size_t> ef_runtime = std::nullopt,
                                         bool enable_partial_results = false) const;           // :313-347
 protected:
  Status ResizeIfFull() override;                                                               // :238-271
 private:
  VectorHNSW(const HnswParams &p, vk_index *ix)
      : VectorBase(IndexerType::kHNSW, p.dimensions, p.metric, ix), m_(p.m), ef_construction_(p.ef_construction),
        ef_runtime_(p.ef_runtime), block_size_(p.hnsw_block_size) {}
  uint32_t m_, ef_construction_, ef_runtime_, block_size_;
};

}  // namespace indexes

namespace query {
constexpr double kPrefilteringThresholdRatio = 0.001;                                           // valkey_search_options.cc:367
bool UsePreFiltering(size_t estimated_num_of_keys, const indexes::VectorBase *vector_index);    // planner.cc:21-45
}  // namespace query

}  // namespace vsa
