// vk_algo.h -- the hnswlib-shaped facade INTEGRATION.md section 1 describes, as a compilable header:
// one class with the member names valkey-search's VectorFlat<T> / VectorHNSW<T> call on their `algo_`
// (hnswlib::AlgorithmInterface, third_party/hnswlib/hnswlib.h:215-235, plus the members the two classes
// reach into: setEf, resizeIndex, markDelete, SaveIndex/LoadIndex), implemented on the C ABI of
// vk_index.h.  Header-only and free of valkey-search types so that it compiles on its own (the adaptor
// maps data_model::DistanceMetric to vk_metric at the Create call sites, vector_flat.cc:53-74 /
// vector_hnsw.cc:84-108).  Errors surface as exceptions derived from std::runtime_error because that
// is what the two classes already catch around every hnswlib call (vector_flat.cc:165-176,238-242,
// vector_hnsw.cc:186-197,331-335); the "exceeds the specified limit" text that drives their
// resize-and-retry loop is preserved.
#ifndef VK_ALGO_H_
#define VK_ALGO_H_

#include <cstddef>
#include <cstdint>
#include <optional>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vk_index.h"

namespace vkalgo {

using labeltype = size_t;   // hnswlib::labeltype (hnswlib.h:141)

struct Error : std::runtime_error {
  int code;
  explicit Error(int c) : std::runtime_error(vk_last_error()), code(c) {}
};
inline void Check(int rc) {
  if (rc != VK_OK) throw Error(rc);
}

template <typename dist_t>
class Algo {
 public:
  using ResultQueue = std::priority_queue<std::pair<dist_t, labeltype>>;

  Algo(vk_algo algo, uint32_t dim, vk_metric metric, size_t max_elements, size_t M = 16, size_t ef_construction = 200,
       uint32_t block_size = 1024, bool allow_replace_deleted = false, int device_id = -1) {
    vk_index_params p{};
    p.struct_size = sizeof p;
    p.algo = algo;
    p.metric = metric;
    p.dtype = VK_DTYPE_F32;
    p.dim = dim;
    p.block_size = block_size;
    p.initial_cap = max_elements;
    p.m = (uint32_t)M;
    p.ef_construction = (uint32_t)ef_construction;
    p.ef_runtime = 10;                                   // hnswlib's default ef_ (hnswalg.h:131)
    p.allow_replace_deleted = allow_replace_deleted ? 1u : 0u;
    p.random_seed = 100;
    p.device_id = device_id;
    params_ = p;
    Check(vk_index_create(&p, &ix_));
  }
  // LoadIndex (bruteforce.h:171-207 / hnswalg.h:887-1139): read_chunk pulls one RDB chunk per call
  Algo(const vk_index_params &p, vk_read_chunk_fn read_chunk, void *user) : params_(p) {
    Check(vk_index_load(&p, read_chunk, user, &ix_));
  }
  ~Algo() { vk_index_destroy(ix_); }
  Algo(const Algo &) = delete;
  Algo &operator=(const Algo &) = delete;

  // ---- hnswlib::AlgorithmInterface ----------------------------------------------------------------
  void addPoint(const void *datapoint, labeltype label, bool /*replace_deleted*/ = false) {
    Check(vk_index_add(ix_, label, datapoint));
  }
  // the whole (internal_id, vector) set of one backfill step in one call (bulk ingest)
  void addPoints(const uint64_t *labels, const void *rows, size_t n) { Check(vk_index_add_batch(ix_, labels, rows, n)); }
  void removePoint(labeltype label) { Check(vk_index_remove(ix_, label)); }   // BruteforceSearch
  void markDelete(labeltype label) { Check(vk_index_remove(ix_, label)); }    // HierarchicalNSW
  void resizeIndex(size_t new_max_elements) { Check(vk_index_resize(ix_, new_max_elements)); }
  void setEf(size_t ef) { Check(vk_index_set_ef(ix_, (uint32_t)ef)); }

  // searchKnn(query, k[, ef], filter, canceller): the filter arrives as an allow-bitmap over labels and
  // the canceller as a flag; the result is the max-heap VectorBase::CreateReply pops
  // (vector_base.cc:258-277)
  ResultQueue searchKnn(const void *query, size_t k, std::optional<size_t> ef = std::nullopt,
                        const uint64_t *allow_bits = nullptr, uint64_t allow_nbits = 0,
                        const volatile int *cancel_flag = nullptr, bool partial_ok = true) const {
    std::vector<float> d(k ? k : 1);
    std::vector<uint64_t> l(k ? k : 1);
    uint64_t n = 0;
    Check(vk_index_search(ix_, query, k, ef.value_or(0), allow_bits, allow_nbits, cancel_flag, partial_ok ? 1 : 0,
                          d.data(), l.data(), &n));
    ResultQueue out;
    for (uint64_t i = 0; i < n; ++i) out.emplace((dist_t)d[i], (labeltype)l[i]);
    return out;
  }
  // pre-filtered exact kNN over an explicit label list (AddPrefilteredKey, vector_base.cc:509-530)
  ResultQueue searchLabels(const void *query, size_t k, const uint64_t *labels, size_t n_labels) const {
    std::vector<float> d(k ? k : 1);
    std::vector<uint64_t> l(k ? k : 1);
    uint64_t n = 0;
    Check(vk_index_search_labels(ix_, query, k, labels, n_labels, d.data(), l.data(), &n));
    ResultQueue out;
    for (uint64_t i = 0; i < n; ++i) out.emplace((dist_t)d[i], (labeltype)l[i]);
    return out;
  }
  // fstdistfunc_(query, getDataByLabel(label)) as ComputeDistanceFromRecordImpl uses it
  dist_t distance(labeltype label, const void *query) const {
    float d = 0;
    Check(vk_index_distance(ix_, label, query, &d));
    return (dist_t)d;
  }
  void getDataByLabel(labeltype label, void *out_row) const { Check(vk_index_get_row(ix_, label, out_row)); }

  // SaveIndex(RDBChunkOutputStream&): write_chunk forwards each chunk to SaveChunk
  void SaveIndex(vk_write_chunk_fn write_chunk, void *user) const { Check(vk_index_save(ix_, write_chunk, user)); }

  // the fields valkey-search reads off the object
  size_t getCurrentElementCount() const { return stats().count; }
  size_t getDeletedCount() const { return stats().deleted; }
  size_t getMaxElements() const { return stats().capacity; }                  // GetCapacity()
  vk_index_stats stats() const {
    vk_index_stats s;
    Check(vk_index_get_stats(ix_, &s));
    return s;
  }
  void flush() { Check(vk_index_flush(ix_)); }                                // write -> read phase switch
  void setCoalescing(uint32_t max_batch, uint32_t max_wait_us) { Check(vk_index_set_coalescing(ix_, max_batch, max_wait_us)); }
  vk_index *handle() const { return ix_; }
  const vk_index_params &params() const { return params_; }

 private:
  vk_index *ix_ = nullptr;
  vk_index_params params_{};
};

}  // namespace vkalgo
#endif  // VK_ALGO_H_
