// filter_set.hpp -- a device-resident allow-set: the materialised BaseFilterFunctor (third_party/hnswlib/hnswlib.h:144-149)
// of one predicate, built ONCE and shared by every search that carries it.
//
// The reference builds an InlineVectorFilter per FT.SEARCH (src/query/search.cc:135-145) and calls it per visited candidate.
// Here a filter is a bitmap over the labels in HBM (one copy per device of the index), built on the device from the id
// lists / id runs the predicate's EntriesFetchers yield (filter_build.hip) or uploaded from a host bitmap, reference
// counted (a search in flight keeps it alive: its caller may drop the handle, or leave on a timeout, at any time) and
// cacheable under a caller-chosen key and epoch (vk_index_filter_cache_*): a repeated `@tag:{x}` costs a hash lookup.
#pragma once
#include <stdint.h>

#include <memory>
#include <vector>

#include "row_store.hpp"   // Status

namespace vk {

class FilterSet {
 public:
  ~FilterSet();
  FilterSet(const FilterSet &) = delete;
  FilterSet &operator=(const FilterSet &) = delete;
  uint64_t nbits() const { return nbits_; }
  uint64_t words() const { return (nbits_ + 63) / 64; }
  uint64_t id() const { return id_; }               // unique in the process, never 0: dispatcher lanes of a FLAT index are keyed by it
  uint64_t allowed() const { return allowed_; }     // set bits (counted on the device)
  uint64_t device_bytes() const { return (words() + 1) * 8 * copies_.size(); }
  const uint64_t *bits_on(int device) const {       // nullptr: no copy on that device
    for (const Copy &c : copies_)
      if (c.device == device) return c.bits;
    return nullptr;
  }
  Status read(uint64_t *out_words, uint64_t n_words) const;   // the bitmap back on the host (tests, save)
  // ids: labels in any order, duplicates allowed; runs: [n_runs][2] = first, last (inclusive); host_bits: a bitmap of
  // nbits bits to start from (nullptr = empty).  Labels >= nbits are ignored.  Blocks until every copy is complete.
  static Status build(const std::vector<int> &devices, uint64_t nbits, const uint64_t *ids, uint64_t n_ids, const uint64_t *runs,
                      uint64_t n_runs, const uint64_t *host_bits, std::shared_ptr<FilterSet> *out);
  // dst = a OP b (0 and, 1 or, 2 and-not) on every device both live on; the two must have the same nbits (else an error)
  static Status combine(const FilterSet &a, const FilterSet &b, uint32_t op, std::shared_ptr<FilterSet> *out);
  // n combinations, one launch and one wait per device: out[i] = a[i] OP[i] b[i] (all of one index and one nbits)
  static Status combine_batch(const FilterSet *const *a, const FilterSet *const *b, const uint32_t *ops, uint64_t n,
                              std::vector<std::shared_ptr<FilterSet>> *out);

 private:
  FilterSet() = default;
  static Status allocate(const std::vector<int> &devices, uint64_t nbits, std::shared_ptr<FilterSet> *out);
  struct Copy { int device; uint64_t *bits; };
  std::vector<Copy> copies_;
  uint64_t nbits_ = 0, id_ = 0, allowed_ = 0;
};

}  // namespace vk
