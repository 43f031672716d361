// hnsw_graph.hpp -- host-side HNSW graph: the authoritative copy of what
// hnswlib::HierarchicalNSW<float> holds (third_party/hnswlib/hnswalg.h), built on the host
// exactly as the reference builds it (concurrent addPoint from writer threads,
// hnswalg.h:1523-1650) and mirrored to HBM for the device search (hnsw_search.hip).
//
// Same ALGORITHM as the reference's builder (the HNSW paper's: beam search per layer, diversity heuristic, mutual links
// with re-selection of full lists, two-hop repair on update) and the same seeded level generator
// (std::default_random_engine seeded 100), organised around this file's own structures: candidate sets are vectors kept
// nearest first under a total (distance, id) order, not std::priority_queues compared on the distance alone, so what
// happens on equal distances is defined here and not by libstdc++'s heap; on tie-free data a single-threaded insert
// sequence produces the graph hnswlib produces, link for link (tests/test_host_graph.py, against the oracle).  Storage is
// flat fixed-stride arrays instead of hnswlib's ChunkedArray of {links | pointer | label} records so the level-0 table can be
// uploaded as is: links0[id][0] = count (low 16 bits) | tombstone (bit 16, hnswalg.h:1259-1262),
// links0[id][1..maxM0] = neighbour ids.
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "host_dist.hpp"
#include "row_store.hpp"

namespace vk {

#ifdef VK_PROFILE_LOCKS
inline std::atomic<uint64_t> g_spin_wait_cycles{0}, g_spin_waits{0};
#endif

class HnswGraph {
 public:
  static constexpr uint32_t kDeleteFlag = 0x00010000u;
  static constexpr uint32_t kNone = 0xFFFFFFFFu;
  enum Dirty : uint8_t { kDirtyL0 = 1, kDirtyUpper = 2 };

  HnswGraph(uint32_t dim, bool l2, size_t max_elements, size_t M, size_t ef_construction, size_t seed,
            bool allow_replace_deleted);
  ~HnswGraph();

  // addPoint(data, label, replace_deleted = allow_replace_deleted_) hnswalg.h:1278-1340.
  // Thread safe.  *out_id = internal id that now holds the row; *row_changed = the
  // device copy of that slot's row/label must be (re)written.
  Status add(const float *row, uint64_t label, uint32_t *out_id);
  Status mark_delete(uint64_t label);       // hnswalg.h:1173-1187
  Status resize(size_t new_max);            // hnswalg.h:758-777; caller excludes other calls
  void set_ef(size_t ef) { ef_ = ef; }

  size_t dim() const { return dim_; }
  size_t M() const { return M_; }
  size_t maxM() const { return maxM_; }
  size_t maxM0() const { return maxM0_; }
  size_t ef() const { return ef_; }
  size_t ef_construction() const { return efC_; }
  size_t max_elements() const { return max_elements_; }
  size_t count() const { return count_.load(std::memory_order_acquire); }
  size_t deleted_count() const { return num_deleted_.load(); }
  int max_level() const { return maxlevel_; }
  uint32_t entry_point() const { return enterpoint_; }
  double mult() const { return mult_; }

  bool lookup(uint64_t label, uint32_t *id) const;      // label_lookup_
  bool is_deleted(uint32_t id) const { return (__atomic_load_n(links0(id), __ATOMIC_RELAXED) & kDeleteFlag) != 0; }
  uint64_t label_of(uint32_t id) const { return labels_[id]; }
  int level_of(uint32_t id) const { return levels_[id]; }
  const float *row(uint32_t id) const { return rows_[id >> kChunkShift] + (size_t)(id & kChunkMask) * dim_; }
  const uint32_t *links0(uint32_t id) const { return l0_.get() + (size_t)id * (maxM0_ + 1); }
  const uint32_t *upper(uint32_t id, int level) const { return upper_[id] + (size_t)(level - 1) * (maxM_ + 1); }
  uint32_t upper_slot(uint32_t id) const { return upper_slot_[id]; }   // first device slot of id's upper lists
  uint32_t upper_slots_used() const { return upper_slots_used_.load(); }
  uint64_t max_label() const;                                          // VectorHNSW::GetMaxInternalLabel: the largest label ever held
  void note_label(uint64_t label) {
    uint64_t cur = max_label_.load(std::memory_order_relaxed);
    while (label > cur && !max_label_.compare_exchange_weak(cur, label, std::memory_order_relaxed)) {}
  }
  uint64_t host_bytes() const;

  // dirty tracking for the device mirror
  uint8_t take_dirty(uint32_t id) { return dirty_[id].exchange(0, std::memory_order_acq_rel); }
  bool any_dirty() const { return any_dirty_.load(std::memory_order_acquire); }
  void clear_any_dirty() { any_dirty_.store(false, std::memory_order_release); }
  // ids whose dirty byte went from 0 to non-zero since the last call (flush walks these, not the
  // whole table: a scan of 10M atomics per flush was most of the cost of a bulk build)
  std::vector<uint32_t> take_dirty_ids() {
    std::lock_guard<std::mutex> g(dirty_mu_);
    std::vector<uint32_t> out;
    out.swap(dirty_ids_);
    return out;
  }

  // device-assisted bulk insert (hnsw_build.hip drives level 0): the caller holds the index
  // exclusively.  bulk_register = the slot/label/level bookkeeping of addPoint (:1523-1583) for n new
  // labels (levels drawn in order from the same generator); bulk_link_upper = the rest of addPoint
  // restricted to levels >= 1, thread safe like add().
  bool bulk_possible(const uint64_t *labels, size_t n) const;
  Status bulk_register(const float *rows, const uint64_t *labels, size_t n, uint32_t *first_id);
  Status bulk_link_upper(uint32_t id);
  uint32_t *links0_table() { return l0_.get(); }

  // load path (persist): install a fully formed element
  Status load_element(uint32_t id, const uint32_t *links0_words, const float *row, uint64_t label);
  Status load_upper(uint32_t id, const uint32_t *words, size_t n_words);
  Status load_labels(size_t count);   // rebuild label_lookup_ with the duplicate-label rule (:1033-1052)
  void load_finish(size_t count, int maxlevel, uint32_t enterpoint);

 private:
  static constexpr uint32_t kChunkShift = 10, kChunkMask = (1u << kChunkShift) - 1;
  // a neighbour candidate; sets of them are kept nearest first under a TOTAL order (distance, then id)
  struct Near { float d; uint32_t id; };
  static bool nearer(const Near &a, const Near &b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }
  using NearList = std::vector<Near>;

  struct Spin {
    std::atomic_flag &f;
    explicit Spin(std::atomic_flag &fl) : f(fl) {
      // an in-flight insert holds its own node lock for its whole duration (hnswalg.h:1561):
      // back off to the scheduler instead of burning the core
#ifdef VK_PROFILE_LOCKS
      if (!f.test_and_set(std::memory_order_acquire)) return;
      const uint64_t t0 = __builtin_ia32_rdtsc();
      for (unsigned spins = 0; f.test_and_set(std::memory_order_acquire); ++spins) {
        if (spins < 64) __builtin_ia32_pause();
        else std::this_thread::yield();
      }
      g_spin_wait_cycles.fetch_add(__builtin_ia32_rdtsc() - t0, std::memory_order_relaxed);
      g_spin_waits.fetch_add(1, std::memory_order_relaxed);
#else
      for (unsigned spins = 0; f.test_and_set(std::memory_order_acquire); ++spins) {
        if (spins < 64) __builtin_ia32_pause();
        else std::this_thread::yield();
      }
#endif
    }
    ~Spin() { f.clear(std::memory_order_release); }
  };
  // "seen in this search" without clearing: a node is seen when its stamp equals the search's epoch
  struct SeenStamps {
    uint16_t epoch = 0;
    std::vector<uint16_t> stamp;
    explicit SeenStamps(size_t n) : stamp(n, 0) {}
    uint16_t advance() {
      if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0); epoch = 1; }
      return epoch;
    }
  };

  float dist(const float *a, const float *b) const { return dist_(a, b, dim_); }
  uint32_t *links0_mut(uint32_t id) { return l0_.get() + (size_t)id * (maxM0_ + 1); }
  uint32_t *upper_mut(uint32_t id, int level) { return upper_[id] + (size_t)(level - 1) * (maxM_ + 1); }
  uint32_t *list_at(uint32_t id, int level) { return level == 0 ? links0_mut(id) : upper_mut(id, level); }
  // (an atomic load: mark_delete sets the tombstone bit of the same word while searches read the count)
  static unsigned list_count(const uint32_t *ll) { return __atomic_load_n(ll, __ATOMIC_RELAXED) & 0xFFFFu; }
  // word 0 of a level-0 list carries the neighbour count (low 16 bits, written under the node's link lock) AND the
  // tombstone bit (written by markDelete under the label lock only): both sides update it atomically, so neither a
  // count nor a tombstone can be lost when a remove overlaps an insert that re-links the node (hnswlib keeps them in
  // separate bytes, hnswalg.h:1196,1269)
  static void set_list_count(uint32_t *ll, unsigned n) {
    uint32_t old = __atomic_load_n(ll, __ATOMIC_RELAXED);
    while (!__atomic_compare_exchange_n(ll, &old, (old & 0xFFFF0000u) | (n & 0xFFFFu), true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
  }
  float *row_mut(uint32_t id) { return rows_[id >> kChunkShift] + (size_t)(id & kChunkMask) * dim_; }
  Status ensure_row_chunk(uint32_t id);
  void mark(uint32_t id, int level) {
    const uint8_t prev = dirty_[id].fetch_or(level == 0 ? kDirtyL0 : kDirtyUpper, std::memory_order_acq_rel);
    if (prev == 0) {
      std::lock_guard<std::mutex> g(dirty_mu_);
      dirty_ids_.push_back(id);
    }
    any_dirty_.store(true, std::memory_order_release);
  }
  std::unique_ptr<SeenStamps> borrow_seen();
  void return_seen(std::unique_ptr<SeenStamps> v);
  int draw_level();

  // the builder (hnsw_graph.cc): what each stands in for is cited there
  NearList beam_search(uint32_t entry, const float *q, int layer);
  void keep_diverse(NearList &cands, size_t keep) const;
  static void store_list(uint32_t *list, const NearList &nearest_first);
  Status wire(uint32_t node, int layer, NearList &cands, bool rewire, uint32_t *closest);
  uint32_t descend(const float *q, uint32_t from, int top, int stop);
  Status link_layers(uint32_t node, const float *q, uint32_t start, int first, int last, uint32_t entry, bool rewire);
  Status insert(const float *row, uint64_t label, uint32_t *out_id);
  Status refresh(const float *row, uint32_t id, float share);
  Status set_tombstone(uint32_t id);
  Status clear_tombstone(uint32_t id);
  std::vector<uint32_t> snapshot_list(uint32_t id, int level);
  void alloc_tables(size_t n, size_t keep);

  size_t dim_;
  host_dist_fn dist_;
  size_t max_elements_;
  std::atomic<size_t> count_{0};
  std::atomic<size_t> num_deleted_{0};
  std::atomic<uint64_t> max_label_{0};
  size_t M_, maxM_, maxM0_, efC_, ef_ = 10;
  double mult_;
  int maxlevel_ = -1;
  uint32_t enterpoint_ = kNone;
  bool allow_replace_deleted_;

  std::unique_ptr<uint32_t[]> l0_;
  std::unique_ptr<uint32_t *[]> upper_;
  std::unique_ptr<uint32_t[]> upper_slot_;
  std::atomic<uint32_t> upper_slots_used_{0};
  std::unique_ptr<int[]> levels_;
  std::unique_ptr<uint64_t[]> labels_;
  std::vector<float *> rows_;
  std::mutex rows_mu_;
  std::unique_ptr<std::atomic_flag[]> link_locks_;
  std::unique_ptr<std::atomic<uint8_t>[]> dirty_;
  std::atomic<bool> any_dirty_{false};
  std::mutex dirty_mu_;
  std::vector<uint32_t> dirty_ids_;

  std::mutex global_;
  mutable std::mutex label_lookup_lock_;
  std::unordered_map<uint64_t, uint32_t> label_lookup_;
  static constexpr size_t kLabelLocks = 65536;
  std::unique_ptr<std::mutex[]> label_op_locks_;
  std::mutex vacant_mu_;
  std::unordered_set<uint32_t> vacant_;          // tombstoned slots a new label may take over (allow_replace_deleted)
  std::mutex rng_mu_;
  std::default_random_engine level_rng_, refresh_rng_;
  std::mutex seen_mu_;
  std::vector<std::unique_ptr<SeenStamps>> seen_pool_;
};

}  // namespace vk
