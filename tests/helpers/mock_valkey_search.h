// mock_valkey_search.h -- TEST INFRASTRUCTURE: the smallest set of declarations under which include/vk_vector_adaptor.h
// compiles OUTSIDE the valkey-search tree (tests/test_abi_symbols.py builds tests/helpers/adaptor_check.cc against it with
// -Wall -Werror).  It is a MOCK of the names the adaptor touches, written for this check only -- not a copy of the module's
// headers and not a build of the reference:
//   * the pure virtuals of valkey_search::indexes::VectorBase that VectorFlat<T> / VectorHNSW<T> override
//     (src/indexes/vector_base.h:129-282 -- names, parameter types, const-ness), so that a signature drift between the
//     adaptor and the interface it must satisfy is a compile error here;
//   * stand-ins for the abseil / vmsdk / hnswlib / protobuf types that appear in those signatures, with just the members
//     the adaptor calls.
// In the module, VK_ADAPTOR_IN_TREE is defined and the adaptor includes the real headers instead.
#ifndef MOCK_VALKEY_SEARCH_H_
#define MOCK_VALKEY_SEARCH_H_

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <variant>
#include <vector>

namespace absl {
using string_view = std::string_view;
enum class StatusCode { kOk = 0, kCancelled = 1, kInvalidArgument = 3, kNotFound = 5, kResourceExhausted = 8, kFailedPrecondition = 9, kInternal = 13, kUnavailable = 14 };
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
inline Status InternalError(std::string_view m) { return Status(StatusCode::kInternal, std::string(m)); }
inline Status InvalidArgumentError(std::string_view m) { return Status(StatusCode::kInvalidArgument, std::string(m)); }
inline Status NotFoundError(std::string_view m) { return Status(StatusCode::kNotFound, std::string(m)); }
inline Status CancelledError(std::string_view m) { return Status(StatusCode::kCancelled, std::string(m)); }
inline Status ResourceExhaustedError(std::string_view m) { return Status(StatusCode::kResourceExhausted, std::string(m)); }
inline Status UnavailableError(std::string_view m) { return Status(StatusCode::kUnavailable, std::string(m)); }
inline Status FailedPreconditionError(std::string_view m) { return Status(StatusCode::kFailedPrecondition, std::string(m)); }
template <typename T>
class StatusOr {
 public:
  StatusOr(const Status &s) : v_(s) {}                       // NOLINT: implicit like absl's
  StatusOr(T v) : v_(std::move(v)) {}                        // NOLINT
  bool ok() const { return v_.index() == 1; }
  const Status &status() const { static const Status k; return ok() ? k : std::get<0>(v_); }
  T &value() { return std::get<1>(v_); }
  const T &value() const { return std::get<1>(v_); }
  T &operator*() { return value(); }

 private:
  std::variant<Status, T> v_;
};
}  // namespace absl

struct ValkeyModuleCtx;
extern "C" {
// the reply calls RespondWithInfoImpl makes (vmsdk/src/valkey_module_api/valkey_module.h); defined by the check program
int ValkeyModule_ReplyWithSimpleString(ValkeyModuleCtx *ctx, const char *msg);
int ValkeyModule_ReplyWithLongLong(ValkeyModuleCtx *ctx, long long ll);
}

namespace hnswlib {
using labeltype = size_t;                                    // hnswlib.h:141
class BaseFilterFunctor {                                    // hnswlib.h:144-149
 public:
  virtual bool operator()(labeltype) { return true; }
  virtual ~BaseFilterFunctor() = default;
};
}  // namespace hnswlib

namespace valkey_search {
class InternedString {
 public:
  explicit InternedString(std::string s) : s_(std::move(s)) {}
  absl::string_view Str() const { return s_; }

 private:
  std::string s_;
};
using InternedStringPtr = std::shared_ptr<InternedString>;

namespace cancel {                                           // src/utils/cancel.h
struct Base {
  virtual ~Base() = default;
  virtual bool IsCancelled() = 0;
  virtual void Cancel() = 0;
};
using Token = std::shared_ptr<Base>;
}  // namespace cancel

namespace query { inline constexpr absl::string_view kTimeoutMsg = "Search operation cancelled due to timeout"; }

namespace data_model {
enum DistanceMetric { DISTANCE_METRIC_UNSPECIFIED = 0, DISTANCE_METRIC_L2 = 1, DISTANCE_METRIC_IP = 2, DISTANCE_METRIC_COSINE = 3 };
enum AttributeDataType { ATTRIBUTE_DATA_TYPE_UNSPECIFIED = 0, ATTRIBUTE_DATA_TYPE_HASH = 1, ATTRIBUTE_DATA_TYPE_JSON = 2 };
struct HNSWAlgorithm {                                       // index_schema.proto HNSWAlgorithm
  uint32_t m_ = 16, ef_construction_ = 200, ef_runtime_ = 10;
  uint32_t m() const { return m_; }
  uint32_t ef_construction() const { return ef_construction_; }
  uint32_t ef_runtime() const { return ef_runtime_; }
  void set_m(uint32_t v) { m_ = v; }
  void set_ef_construction(uint32_t v) { ef_construction_ = v; }
  void set_ef_runtime(uint32_t v) { ef_runtime_ = v; }
};
struct FlatAlgorithm {
  uint32_t block_size_ = 1024;
  uint32_t block_size() const { return block_size_; }
  void set_block_size(uint32_t v) { block_size_ = v; }
};
struct VectorIndex {                                         // index_schema.proto VectorIndex
  uint32_t dimension_count_ = 0, initial_cap_ = 1024;
  DistanceMetric distance_metric_ = DISTANCE_METRIC_L2;
  HNSWAlgorithm hnsw_;
  FlatAlgorithm flat_;
  uint32_t dimension_count() const { return dimension_count_; }
  uint32_t initial_cap() const { return initial_cap_; }
  DistanceMetric distance_metric() const { return distance_metric_; }
  const HNSWAlgorithm &hnsw_algorithm() const { return hnsw_; }
  const FlatAlgorithm &flat_algorithm() const { return flat_; }
  HNSWAlgorithm *mutable_hnsw_algorithm() { return &hnsw_; }
  FlatAlgorithm *mutable_flat_algorithm() { return &flat_; }
};
}  // namespace data_model

// rdb_serialization.h:289-340 -- one chunk per call
class RDBChunkOutputStream {
 public:
  virtual ~RDBChunkOutputStream() = default;
  virtual absl::Status SaveChunk(const char *data, size_t len) = 0;
};
// rdb_serialization.h:162-205: the chunks of one supplemental-content section, move-only; the mock iterates a vector
class SupplementalContentChunkIter {
 public:
  explicit SupplementalContentChunkIter(std::vector<std::string> chunks) : chunks_(std::move(chunks)) {}
  SupplementalContentChunkIter(SupplementalContentChunkIter &&) noexcept = default;
  SupplementalContentChunkIter &operator=(SupplementalContentChunkIter &&) noexcept = default;
  SupplementalContentChunkIter(const SupplementalContentChunkIter &) = delete;
  bool HasNext() const { return next_ < chunks_.size(); }
  absl::StatusOr<std::unique_ptr<std::string>> Next() { return std::make_unique<std::string>(std::move(chunks_[next_++])); }

 private:
  std::vector<std::string> chunks_;
  size_t next_ = 0;
};
// rdb_serialization.h:289-303, rdb_serialization.cc:103-109
class RDBChunkInputStream {
 public:
  explicit RDBChunkInputStream(SupplementalContentChunkIter &&iter) : iter_(std::move(iter)) {}
  RDBChunkInputStream(RDBChunkInputStream &&) noexcept = default;
  absl::StatusOr<std::unique_ptr<std::string>> LoadChunk() {
    if (!iter_.HasNext()) return absl::NotFoundError("No more elements remaining");
    return iter_.Next();
  }

 private:
  SupplementalContentChunkIter iter_;
};
// attribute_data_type.h:61-81 (the one member LoadFromRDB uses)
class AttributeDataType {
 public:
  virtual ~AttributeDataType() = default;
  virtual data_model::AttributeDataType ToProto() const = 0;
};
class HashAttributeDataType : public AttributeDataType {
 public:
  data_model::AttributeDataType ToProto() const override { return data_model::ATTRIBUTE_DATA_TYPE_HASH; }
};

namespace indexes {
enum class IndexerType { kHNSW, kFlat };
// index_base.h:103-116: what Tag::Search / Numeric::Search hand to the query layer
class EntriesFetcherIteratorBase {
 public:
  virtual bool Done() const = 0;
  virtual void Next() = 0;
  virtual const InternedStringPtr &operator*() const = 0;
  virtual ~EntriesFetcherIteratorBase() = default;
};
class EntriesFetcherBase {
 public:
  virtual size_t Size() const = 0;
  virtual ~EntriesFetcherBase() = default;
  virtual std::unique_ptr<EntriesFetcherIteratorBase> Begin() = 0;
};
struct Neighbor {
  InternedStringPtr external_id;
  float distance = 0.f;
  Neighbor() = default;
  Neighbor(InternedStringPtr id, float d) : external_id(std::move(id)), distance(d) {}
};

// The slice of VectorBase (src/indexes/vector_base.h:129-282) a backend class implements.  The key <-> internal-id maps,
// normalisation and CreateReply live in the real base class; the mock keeps just enough of them for Search() to compile.
class VectorBase {
 public:
  virtual ~VectorBase() = default;
  virtual size_t GetCapacity() const = 0;
  virtual size_t GetDataTypeSize() const = 0;
  virtual uint64_t GetMaxInternalLabel() const { return 0; }
  virtual size_t GetLabelCount() const { return 0; }
  bool GetNormalize() const { return normalize_; }
  int GetVectorDataSize() const { return (int)GetDataTypeSize() * dimensions_; }
  // (the mock's key of internal id N is the string "N"; ids that were removed have no key any more)
  absl::StatusOr<InternedStringPtr> GetKeyDuringSearch(uint64_t internal_id) const {
    if (!mock_all_live_) {
      std::lock_guard<std::mutex> l(mock_mu_);
      if (!mock_live_.count(internal_id)) return absl::InvalidArgumentError("Record was not found");
    }
    // The real class FINDS the interned key (key_by_internal_id_.find, vector_base.cc:212-219: a hash lookup and a reference
    // count, no allocation): the mock's keys are made on first use and kept
    KeyShard &sh = mock_keys_[internal_id & (kKeyShards - 1)];
    std::lock_guard<std::mutex> l(sh.mu);
    auto it = sh.map.find(internal_id);
    if (it == sh.map.end()) it = sh.map.emplace(internal_id, std::make_shared<InternedString>(std::to_string(internal_id))).first;
    return it->second;
  }
  void MockAllKeysLive() { mock_all_live_ = true; }   // (an adopted index: every label has its key)
  // vector_base.cc:333-338: the VectorTracker entry LoadIndex calls -- intern, then the virtual
  char *TrackVector(uint64_t internal_id, char *vector, size_t len) {
    auto interned = std::make_shared<InternedString>(std::string(vector, len));
    {
      std::lock_guard<std::mutex> l(mock_mu_);
      mock_live_.insert(internal_id);
    }
    TrackVector(internal_id, interned);
    return const_cast<char *>(interned->Str().data());
  }
  // What the real base class's public entry points (AddRecord, RemoveRecord, ModifyRecord, SaveIndex, RespondWithInfo,
  // ToProto, GetValue, ComputeDistanceFromRecord: vector_base.cc) reach after their key <-> id bookkeeping: the Impl
  // virtuals, called here with the internal id directly.
  absl::Status MockAdd(uint64_t id, absl::string_view record, const InternedStringPtr &vector) {
    absl::Status st = AddRecordImpl(id, record);
    if (st.ok()) {
      TrackVector(id, vector);
      std::lock_guard<std::mutex> l(mock_mu_);
      mock_live_.insert(id);
    }
    return st;
  }
  absl::Status MockModify(uint64_t id, absl::string_view record, const InternedStringPtr &vector) {
    absl::Status st = ModifyRecordImpl(id, record);
    if (st.ok()) TrackVector(id, vector);
    return st;
  }
  absl::Status MockRemove(uint64_t id) {
    absl::Status st = RemoveRecordImpl(id);
    if (st.ok()) {
      UnTrackVector(id);
      std::lock_guard<std::mutex> l(mock_mu_);
      mock_live_.erase(id);
    }
    return st;
  }
  bool MockIsVectorMatch(uint64_t id, const InternedStringPtr &vector) { return IsVectorMatch(id, vector); }
  char *MockGetValue(uint64_t id) const { return GetValueImpl(id); }
  absl::StatusOr<std::pair<float, hnswlib::labeltype>> MockDistance(uint64_t id, absl::string_view query) const {
    return ComputeDistanceFromRecordImpl(id, query);
  }
  int MockInfo(ValkeyModuleCtx *ctx) const { return RespondWithInfoImpl(ctx); }
  void MockToProto(data_model::VectorIndex *proto) const { ToProtoImpl(proto); }
  absl::Status MockSave(RDBChunkOutputStream &out) const { return SaveIndexImpl(out); }

 protected:
  VectorBase(IndexerType t, int dimensions, data_model::AttributeDataType adt, absl::string_view attribute_identifier)
      : dimensions_(dimensions), attribute_identifier_(attribute_identifier), attribute_data_type_(adt), indexer_type_(t) {}
  virtual absl::Status AddRecordImpl(uint64_t internal_id, absl::string_view record) = 0;
  virtual absl::Status RemoveRecordImpl(uint64_t internal_id) = 0;
  virtual absl::Status ModifyRecordImpl(uint64_t internal_id, absl::string_view record) = 0;
  virtual int RespondWithInfoImpl(ValkeyModuleCtx *ctx) const = 0;
  virtual void ToProtoImpl(data_model::VectorIndex *vector_index_proto) const = 0;
  virtual absl::Status SaveIndexImpl(RDBChunkOutputStream &chunked_out) const = 0;   // (the real one takes it by value: a move-only wrapper)
  virtual char *GetValueImpl(uint64_t internal_id) const = 0;
  virtual absl::StatusOr<std::pair<float, hnswlib::labeltype>> ComputeDistanceFromRecordImpl(uint64_t internal_id,
                                                                                            absl::string_view query) const = 0;
  virtual void TrackVector(uint64_t internal_id, const InternedStringPtr &vector) = 0;
  virtual bool IsVectorMatch(uint64_t internal_id, const InternedStringPtr &vector) = 0;
  virtual void UnTrackVector(uint64_t internal_id) = 0;
  // vector_base.cc:203-210.  PRIVATE upstream: the one access change the adaptor needs
  absl::StatusOr<uint64_t> GetInternalIdDuringSearch(const InternedStringPtr &key) const {
    const std::string k(key->Str());
    char *end = nullptr;
    const unsigned long long id = std::strtoull(k.c_str(), &end, 10);
    if (end == k.c_str() || *end) return absl::InvalidArgumentError("Record was not found");
    if (!mock_all_live_) {
      std::lock_guard<std::mutex> l(mock_mu_);
      if (!mock_live_.count(id)) return absl::InvalidArgumentError("Record was not found");
    }
    return (uint64_t)id;
  }

  static constexpr uint64_t kKeyShards = 64;
  struct KeyShard {
    std::mutex mu;
    std::unordered_map<uint64_t, InternedStringPtr> map;
  };
  mutable KeyShard mock_keys_[kKeyShards];
  mutable std::mutex mock_mu_;
  std::unordered_set<uint64_t> mock_live_;
  bool mock_all_live_ = false;
  int dimensions_;
  std::string attribute_identifier_;
  bool normalize_{false};
  data_model::AttributeDataType attribute_data_type_;
  data_model::DistanceMetric distance_metric_{data_model::DISTANCE_METRIC_L2};
  IndexerType indexer_type_;
};
}  // namespace indexes
}  // namespace valkey_search
#endif  // MOCK_VALKEY_SEARCH_H_
