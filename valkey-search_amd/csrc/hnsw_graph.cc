// hnsw_graph.cc -- see hnsw_graph.hpp.  Line references are to
// third_party/hnswlib/hnswalg.h of the reference tree.
#include "hnsw_graph.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>

namespace vk {

namespace {
constexpr int kErrInternal = 4, kErrCapacity = 2, kErrNotFound = 3;
}

void HnswGraph::alloc_tables(size_t n, size_t keep) {
  const size_t m = n ? n : 1;
  std::unique_ptr<uint32_t[]> l0(new uint32_t[m * (maxM0_ + 1)]());
  std::unique_ptr<uint32_t *[]> up(new uint32_t *[m]());
  std::unique_ptr<uint32_t[]> us(new uint32_t[m]);
  std::unique_ptr<int[]> lv(new int[m]());
  std::unique_ptr<uint64_t[]> lb(new uint64_t[m]());
  std::unique_ptr<std::atomic_flag[]> lk(new std::atomic_flag[m]);
  std::unique_ptr<std::atomic<uint8_t>[]> dt(new std::atomic<uint8_t>[m]);
  for (size_t i = 0; i < m; ++i) { lk[i].clear(); dt[i].store(0); us[i] = kNone; }
  if (keep) {
    memcpy(l0.get(), l0_.get(), keep * (maxM0_ + 1) * sizeof(uint32_t));
    memcpy(up.get(), upper_.get(), keep * sizeof(uint32_t *));
    memcpy(us.get(), upper_slot_.get(), keep * sizeof(uint32_t));
    memcpy(lv.get(), levels_.get(), keep * sizeof(int));
    memcpy(lb.get(), labels_.get(), keep * sizeof(uint64_t));
    for (size_t i = 0; i < keep; ++i) dt[i].store(dirty_[i].load());
  }
  l0_ = std::move(l0);
  upper_ = std::move(up);
  upper_slot_ = std::move(us);
  levels_ = std::move(lv);
  labels_ = std::move(lb);
  link_locks_ = std::move(lk);
  dirty_ = std::move(dt);
  rows_.resize((m + kChunkMask) >> kChunkShift, nullptr);
}

HnswGraph::HnswGraph(uint32_t dim, bool l2, size_t max_elements, size_t M, size_t ef_construction, size_t seed,
                     bool allow_replace_deleted)
    : dim_(dim), dist_(l2 ? host_distance_l2() : host_distance_ip()), max_elements_(max_elements),
      allow_replace_deleted_(allow_replace_deleted), label_op_locks_(new std::mutex[kLabelLocks]) {
  M_ = M <= 10000 ? M : 10000;                       // :130-141
  maxM_ = M_;
  maxM0_ = M_ * 2;
  efC_ = std::max(ef_construction, M_);              // :146
  level_rng_.seed(seed);                             // :149
  refresh_rng_.seed(seed + 1);
  mult_ = 1 / log(1.0 * M_);                         // :177
  alloc_tables(max_elements_, 0);
}

HnswGraph::~HnswGraph() {
  const size_t n = count_.load();
  for (size_t i = 0; i < n; ++i) delete[] upper_[i];
  for (float *c : rows_) free(c);
}

uint64_t HnswGraph::host_bytes() const {
  uint64_t b = max_elements_ * ((maxM0_ + 1) * 4 + 8 + 4 + 4 + 8 + 2);
  for (float *c : rows_) if (c) b += ((uint64_t)1 << kChunkShift) * dim_ * 4;
  b += (uint64_t)upper_slots_used_.load() * (maxM_ + 1) * 4;
  b += label_lookup_.size() * 32;
  return b;
}

uint64_t HnswGraph::max_label() const { return max_label_.load(std::memory_order_relaxed); }

bool HnswGraph::lookup(uint64_t label, uint32_t *id) const {
  std::lock_guard<std::mutex> lk(label_lookup_lock_);
  auto it = label_lookup_.find(label);
  if (it == label_lookup_.end()) return false;
  *id = it->second;
  return true;
}

Status HnswGraph::ensure_row_chunk(uint32_t id) {
  const size_t c = id >> kChunkShift;
  std::lock_guard<std::mutex> lk(rows_mu_);
  if (!rows_[c]) {
    void *p = nullptr;
    if (posix_memalign(&p, 64, ((size_t)1 << kChunkShift) * dim_ * sizeof(float)) || !p)
      return Status::Err(kErrInternal, "out of host memory for the HNSW row chunks");
    rows_[c] = static_cast<float *>(p);
  }
  return Status::Ok();
}

std::unique_ptr<HnswGraph::SeenStamps> HnswGraph::borrow_seen() {
  {
    std::lock_guard<std::mutex> lk(seen_mu_);
    if (!seen_pool_.empty()) {
      auto v = std::move(seen_pool_.back());
      seen_pool_.pop_back();
      if (v->stamp.size() >= max_elements_) return v;
    }
  }
  return std::make_unique<SeenStamps>(max_elements_ ? max_elements_ : 1);
}
void HnswGraph::return_seen(std::unique_ptr<SeenStamps> v) {
  std::lock_guard<std::mutex> lk(seen_mu_);
  seen_pool_.push_back(std::move(v));
}

int HnswGraph::draw_level() {   // getRandomLevel: floor(-ln U * 1/ln M), U from the seeded engine
  std::uniform_real_distribution<double> unit(0.0, 1.0);
  return (int)(-log(unit(level_rng_)) * mult_);
}

Status HnswGraph::resize(size_t new_max) {
  if (new_max < count_.load()) return Status::Err(kErrInternal, "resize below the number of elements held");
  alloc_tables(new_max, std::min(count_.load(), max_elements_));
  max_elements_ = new_max;
  std::lock_guard<std::mutex> lk(seen_mu_);
  seen_pool_.clear();
  return Status::Ok();
}

std::vector<uint32_t> HnswGraph::snapshot_list(uint32_t id, int level) {
  Spin hold(link_locks_[id]);
  const uint32_t *ll = list_at(id, level);
  return std::vector<uint32_t>(ll + 1, ll + 1 + list_count(ll));
}

// ---- the building blocks -----------------------------------------------------------------------------------------------
// Everything below works on NearList = a vector of (distance, id) kept NEAREST FIRST under the total order `nearer`
// (distance, then id).  hnswlib moves the same sets between std::priority_queues that compare the distance alone
// (hnswalg.h:202-208), so what it does with equal distances is whatever libstdc++'s heap does; on data without ties the two
// produce the same graph link for link (tests/test_host_graph.py compares with the oracle, which does use that heap).  What
// IS shared with the reference is the stored layout -- a node's list holds its neighbours FARTHEST first, because that is
// the order its heap pops them in and the list order decides the order a search looks at them -- and the on-disk format.

// Beam search over one layer (the role of searchBaseLayer, hnswalg.h:255-347): the ef_construction nearest LIVE nodes
// reachable from `entry`, nearest first.  `open` = reached and not expanded yet (a min-heap), `best` = the survivors (a
// max-heap, worst on top); a node is expanded while it can still improve the worst survivor.  Tombstoned nodes are walked
// through but do not survive.
HnswGraph::NearList HnswGraph::beam_search(uint32_t entry, const float *q, int layer) {
  auto seen = borrow_seen();
  uint16_t *stamp = seen->stamp.data();
  const uint16_t epoch = seen->advance();
  const auto worst_on_top = [](const Near &a, const Near &b) { return nearer(a, b); };
  const auto nearest_on_top = [](const Near &a, const Near &b) { return nearer(b, a); };
  NearList best, open;
  best.reserve(efC_ + 1);
  float reach;   // distance of the worst survivor once the beam is full
  if (!is_deleted(entry)) {
    const float d = dist(q, row(entry));
    best.push_back(Near{d, entry});
    open.push_back(Near{d, entry});
    reach = d;
  } else {
    reach = std::numeric_limits<float>::max();
    open.push_back(Near{reach, entry});
  }
  stamp[entry] = epoch;
  std::vector<uint32_t> adj;
  adj.reserve(maxM0_);
  while (!open.empty()) {
    const Near cur = open.front();
    if (cur.d > reach && best.size() == efC_) break;
    std::pop_heap(open.begin(), open.end(), nearest_on_top);
    open.pop_back();
    {
      Spin hold(link_locks_[cur.id]);
      const uint32_t *ll = list_at(cur.id, layer);
      adj.assign(ll + 1, ll + 1 + list_count(ll));
    }
    for (const uint32_t nb : adj) {
      if (stamp[nb] == epoch) continue;
      stamp[nb] = epoch;
      const float d = dist(q, row(nb));
      if (best.size() >= efC_ && !(reach > d)) continue;
      open.push_back(Near{d, nb});
      std::push_heap(open.begin(), open.end(), nearest_on_top);
      if (!is_deleted(nb)) {
        best.push_back(Near{d, nb});
        std::push_heap(best.begin(), best.end(), worst_on_top);
      }
      if (best.size() > efC_) {
        std::pop_heap(best.begin(), best.end(), worst_on_top);
        best.pop_back();
      }
      if (!best.empty()) reach = best.front().d;
    }
  }
  return_seen(std::move(seen));
  std::sort(best.begin(), best.end(), nearer);
  return best;
}

// Neighbour selection (the heuristic of the HNSW paper, getNeighborsByHeuristic2 hnswalg.h:553-594): walk the candidates
// nearest first and keep one only if it is nearer to the base point than to every candidate already kept -- a candidate
// that sits "behind" a kept one is reachable through it.  Fewer candidates than `keep`: all stay.
void HnswGraph::keep_diverse(NearList &cands, size_t keep) const {
  if (cands.size() < keep) return;
  NearList kept;
  kept.reserve(keep);
  for (const Near &c : cands) {
    if (kept.size() >= keep) break;
    bool shadowed = false;
    for (const Near &k : kept)
      if (dist(row(k.id), row(c.id)) < c.d) { shadowed = true; break; }
    if (!shadowed) kept.push_back(c);
  }
  cands.swap(kept);
}

void HnswGraph::store_list(uint32_t *list, const NearList &nearest_first) {
  const size_t n = nearest_first.size();
  set_list_count(list, (unsigned)n);
  for (size_t i = 0; i < n; ++i) list[1 + i] = nearest_first[n - 1 - i].id;   // farthest first (see above)
}

// Give `node` its list on `layer` from the beam's survivors and link it into each chosen neighbour's list, re-selecting a
// neighbour's list that is full (mutuallyConnectNewElement, hnswalg.h:613-756).  rewire = the node already has lists (an
// updated point being re-attached): its own lock is not held by the caller and back-links may exist already.
// *closest = the nearest chosen neighbour: where the next layer down starts.
Status HnswGraph::wire(uint32_t node, int layer, NearList &cands, bool rewire, uint32_t *closest) {
  const size_t cap = layer ? maxM_ : maxM0_;
  keep_diverse(cands, M_);
  if (cands.size() > M_) return Status::Err(kErrInternal, "neighbour selection kept more than M");
  if (cands.empty()) return Status::Err(kErrInternal, "insert found nothing to link to");
  *closest = cands.front().id;
  for (const Near &c : cands)
    if (layer > levels_[c.id]) return Status::Err(kErrInternal, "link to a layer its target does not have");
  {
    std::unique_ptr<Spin> hold;   // (a fresh insert holds its node's lock for its whole duration)
    if (rewire) hold = std::make_unique<Spin>(link_locks_[node]);
    uint32_t *mine = list_at(node, layer);
    if (!rewire && list_count(mine)) return Status::Err(kErrInternal, "a new node already has links");
    store_list(mine, cands);
    mark(node, layer);
  }
  NearList pool;
  for (size_t i = cands.size(); i-- > 0;) {   // (farthest first, the order they were stored in)
    const uint32_t nb = cands[i].id;
    if (nb == node) return Status::Err(kErrInternal, "a node cannot link to itself");
    Spin hold(link_locks_[nb]);
    uint32_t *theirs = list_at(nb, layer);
    const size_t have = list_count(theirs);
    if (have > cap) return Status::Err(kErrInternal, "a link list is longer than its layer allows");
    if (rewire && std::find(theirs + 1, theirs + 1 + have, node) != theirs + 1 + have) continue;
    if (have < cap) {
      theirs[1 + have] = node;
      set_list_count(theirs, (unsigned)(have + 1));
    } else {   // full: the neighbour keeps the diverse subset of its list plus the newcomer, seen from itself
      const float *base = row(nb);
      pool.clear();
      pool.push_back(Near{dist(row(node), base), node});
      for (size_t j = 0; j < have; ++j) pool.push_back(Near{dist(row(theirs[1 + j]), base), theirs[1 + j]});
      std::sort(pool.begin(), pool.end(), nearer);
      keep_diverse(pool, cap);
      store_list(theirs, pool);
    }
    mark(nb, layer);
  }
  return Status::Ok();
}

// Greedy walk on the sparse upper layers: from `from` on layer `top` down to (not including) `stop`, moving to any
// neighbour strictly nearer to q until none is (hnswalg.h:1593-1618 and its two other copies)
uint32_t HnswGraph::descend(const float *q, uint32_t from, int top, int stop) {
  uint32_t at = from;
  float nearest = dist(q, row(at));
  for (int layer = top; layer > stop; --layer) {
    for (bool moved = true; moved;) {
      moved = false;
      for (const uint32_t nb : snapshot_list(at, layer)) {
        const float d = dist(q, row(nb));
        if (d < nearest) { nearest = d; at = nb; moved = true; }
      }
    }
  }
  return at;
}

// Link `node` on layers first .. last (downwards), each layer's beam starting at the nearest neighbour chosen one layer up.
// A tombstoned entry point is offered as a candidate too (it is walked through but never survives the beam, and the graph
// above may hang on it: hnswalg.h:1624-1634).
Status HnswGraph::link_layers(uint32_t node, const float *q, uint32_t start, int first, int last, uint32_t entry, bool rewire) {
  const bool entry_dead = is_deleted(entry);
  uint32_t at = start;
  for (int layer = first; layer >= last; --layer) {
    NearList found = beam_search(at, q, layer);
    if (rewire) {
      found.erase(std::remove_if(found.begin(), found.end(), [&](const Near &n) { return n.id == node; }), found.end());
      if (found.empty()) continue;
    }
    if (entry_dead) {
      const Near e{dist(q, row(entry)), entry};
      found.insert(std::upper_bound(found.begin(), found.end(), e, nearer), e);
      if (found.size() > efC_) found.pop_back();
    }
    VK_TRY(wire(node, layer, found, rewire, &at));
  }
  return Status::Ok();
}

Status HnswGraph::set_tombstone(uint32_t id) {   // markDeletedInternal
  if (is_deleted(id)) return Status::Err(kErrInternal, "the element is deleted already");
  __atomic_fetch_or(links0_mut(id), kDeleteFlag, __ATOMIC_RELAXED);   // (see set_list_count)
  num_deleted_ += 1;
  mark(id, 0);
  if (allow_replace_deleted_) {
    std::lock_guard<std::mutex> lk(vacant_mu_);
    vacant_.insert(id);
  }
  return Status::Ok();
}

Status HnswGraph::clear_tombstone(uint32_t id) {   // unmarkDeletedInternal
  if (!is_deleted(id)) return Status::Err(kErrInternal, "the element is not deleted");
  __atomic_fetch_and(links0_mut(id), ~kDeleteFlag, __ATOMIC_RELAXED);
  num_deleted_ -= 1;
  mark(id, 0);
  if (allow_replace_deleted_) {
    std::lock_guard<std::mutex> lk(vacant_mu_);
    vacant_.erase(id);
  }
  return Status::Ok();
}

Status HnswGraph::mark_delete(uint64_t label) {   // markDelete, hnswalg.h:1173-1187
  std::lock_guard<std::mutex> one_op_per_label(label_op_locks_[label & (kLabelLocks - 1)]);
  uint32_t id;
  if (!lookup(label, &id)) return Status::Err(kErrNotFound, "Label not found");
  return set_tombstone(id);
}

// A point whose vector changed (updatePoint, hnswalg.h:1342-1430, + repairConnectionsForUpdate :1432-1511).  Per layer:
// the lists of (a `share` of) the node's neighbours are re-selected from the node's two-hop neighbourhood -- the node
// moved, so links that went through it may no longer be the diverse ones -- and then the node itself is searched for from
// the top and re-attached like a new point.
Status HnswGraph::refresh(const float *new_row, uint32_t id, float share) {
  memcpy(row_mut(id), new_row, dim_ * sizeof(float));
  const int top = maxlevel_;
  const uint32_t entry = enterpoint_;
  if (entry == id && count_.load() == 1) return Status::Ok();
  const int node_top = levels_[id];
  std::uniform_real_distribution<float> unit(0.0, 1.0);
  std::vector<uint32_t> around, redo;
  NearList pool;
  for (int layer = 0; layer <= node_top; ++layer) {
    const std::vector<uint32_t> ring = snapshot_list(id, layer);
    if (ring.empty()) continue;
    around.assign(1, id);
    redo.clear();
    for (const uint32_t nb : ring) {
      around.push_back(nb);
      float u;
      {
        std::lock_guard<std::mutex> lk(rng_mu_);
        u = unit(refresh_rng_);     // (one draw per neighbour, in list order: the sequence is part of the build's determinism)
      }
      if (u > share) continue;
      redo.push_back(nb);
      const std::vector<uint32_t> second = snapshot_list(nb, layer);
      around.insert(around.end(), second.begin(), second.end());
    }
    std::sort(around.begin(), around.end());
    around.erase(std::unique(around.begin(), around.end()), around.end());
    std::sort(redo.begin(), redo.end());
    redo.erase(std::unique(redo.begin(), redo.end()), redo.end());
    for (const uint32_t nb : redo) {
      const float *base = row(nb);
      pool.clear();
      for (const uint32_t c : around)
        if (c != nb) pool.push_back(Near{dist(base, row(c)), c});
      std::sort(pool.begin(), pool.end(), nearer);
      if (pool.size() > efC_) pool.resize(efC_);
      keep_diverse(pool, layer == 0 ? maxM0_ : maxM_);
      Spin hold(link_locks_[nb]);
      store_list(list_at(nb, layer), pool);
      mark(nb, layer);
    }
  }
  // ... and the node itself
  const float *q = row(id);
  if (node_top > top) return Status::Err(kErrInternal, "an element's level is above the graph's");
  const uint32_t start = node_top < top ? descend(q, entry, top, node_top) : entry;
  return link_layers(id, q, start, node_top, 0, entry, /*rewire=*/true);
}

// A new point, or the same label again (addPoint(data, label, level), hnswalg.h:1523-1650)
Status HnswGraph::insert(const float *new_row, uint64_t label, uint32_t *out_id) {
  uint32_t node = 0;
  {
    std::unique_lock<std::mutex> table(label_lookup_lock_);
    auto known = label_lookup_.find(label);
    if (known != label_lookup_.end()) {   // the label exists: an in-place update
      const uint32_t existing = known->second;
      if (allow_replace_deleted_ && is_deleted(existing))
        return Status::Err(kErrInternal, "a deleted element cannot be updated in place while deleted slots are being reused");
      table.unlock();
      if (is_deleted(existing)) VK_TRY(clear_tombstone(existing));
      *out_id = existing;
      return refresh(new_row, existing, 1.0f);
    }
    if (count_.load() >= max_elements_)
      return Status::Err(kErrCapacity, "The number of elements exceeds the specified limit");   // (the text callers match: vk_index.h)
    node = (uint32_t)count_.load();
    VK_TRY(ensure_row_chunk(node));
    // the slot is complete before count_ / the label map make it reachable
    memset(links0_mut(node), 0, (maxM0_ + 1) * sizeof(uint32_t));
    labels_[node] = label;
    memcpy(row_mut(node), new_row, dim_ * sizeof(float));
    count_.fetch_add(1, std::memory_order_release);
    label_lookup_[label] = node;
    note_label(label);
  }
  *out_id = node;

  // an insert that raises the graph's top level keeps every other insert out until it is done
  std::unique_lock<std::mutex> raising(global_);
  const int top = maxlevel_;
  Spin mine(link_locks_[node]);
  const int node_top = draw_level();
  if (node_top <= top) raising.unlock();
  levels_[node] = node_top;
  const uint32_t entry = enterpoint_;
  delete[] upper_[node];
  upper_[node] = nullptr;
  if (node_top) {
    upper_[node] = new uint32_t[(size_t)node_top * (maxM_ + 1)]();
    upper_slot_[node] = upper_slots_used_.fetch_add((uint32_t)node_top);
    mark(node, 1);
  }
  mark(node, 0);
  if (entry == kNone) {   // the first point
    enterpoint_ = 0;
    maxlevel_ = node_top;
  } else {
    const uint32_t start = node_top < top ? descend(new_row, entry, top, node_top) : entry;
    VK_TRY(link_layers(node, new_row, start, std::min(node_top, top), 0, entry, /*rewire=*/false));
  }
  if (node_top > top) {
    enterpoint_ = node;
    maxlevel_ = node_top;
  }
  return Status::Ok();
}

// addPoint(data, label, replace_deleted) hnswalg.h:1278-1340: with reuse of deleted slots switched on, a new label takes
// over a tombstoned slot (and is then linked like an update of it)
Status HnswGraph::add(const float *new_row, uint64_t label, uint32_t *out_id) {
  std::lock_guard<std::mutex> one_op_per_label(label_op_locks_[label & (kLabelLocks - 1)]);
  if (!allow_replace_deleted_) return insert(new_row, label, out_id);
  uint32_t existing;
  if (lookup(label, &existing)) {
    if (is_deleted(existing)) {
      {
        std::lock_guard<std::mutex> lk(vacant_mu_);
        vacant_.erase(existing);
      }
      VK_TRY(clear_tombstone(existing));
    }
    *out_id = existing;
    return refresh(new_row, existing, 1.0f);
  }
  uint32_t slot = kNone;
  {
    std::lock_guard<std::mutex> lk(vacant_mu_);
    if (!vacant_.empty()) {
      slot = *vacant_.begin();
      vacant_.erase(vacant_.begin());
    }
  }
  if (slot == kNone) return insert(new_row, label, out_id);
  const uint64_t old_label = labels_[slot];
  labels_[slot] = label;
  {
    std::lock_guard<std::mutex> lk(label_lookup_lock_);
    label_lookup_.erase(old_label);
    label_lookup_[label] = slot;
    note_label(label);
  }
  VK_TRY(clear_tombstone(slot));
  *out_id = slot;
  return refresh(new_row, slot, 1.0f);
}

// ---- device-assisted bulk insert -----------------------------------------------------------------
bool HnswGraph::bulk_possible(const uint64_t *labels, size_t n) const {
  if (count_.load() + n > max_elements_) return false;
  if (allow_replace_deleted_ && num_deleted_.load()) return false;   // those inserts reuse tombstoned slots
  std::lock_guard<std::mutex> lk(label_lookup_lock_);
  std::unordered_set<uint64_t> seen;
  seen.reserve(n * 2);
  for (size_t i = 0; i < n; ++i) {
    const uint64_t l = labels ? labels[i] : (uint64_t)i;
    if (label_lookup_.count(l)) return false;       // an update, not an insert
    if (!seen.insert(l).second) return false;       // the same label twice in one batch: the second is an update
  }
  return true;
}

Status HnswGraph::bulk_register(const float *rows, const uint64_t *labels, size_t n, uint32_t *first_id) {
  std::lock_guard<std::mutex> lock_table(label_lookup_lock_);
  if (count_.load() + n > max_elements_)
    return Status::Err(kErrCapacity, "The number of elements exceeds the specified limit");
  const uint32_t first = (uint32_t)count_.load();
  *first_id = first;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t id = first + (uint32_t)i;
    VK_TRY(ensure_row_chunk(id));
    memset(links0_mut(id), 0, (maxM0_ + 1) * sizeof(uint32_t));
    labels_[id] = labels[i];
    memcpy(row_mut(id), rows + i * dim_, dim_ * sizeof(float));
    const int lv = draw_level();
    levels_[id] = lv;
    delete[] upper_[id];
    upper_[id] = nullptr;
    if (lv) {
      upper_[id] = new uint32_t[(size_t)lv * (maxM_ + 1)]();
      upper_slot_[id] = upper_slots_used_.fetch_add((uint32_t)lv);
      mark(id, 1);
    }
    mark(id, 0);
    label_lookup_[labels[i]] = id;
    note_label(labels[i]);
  }
  count_.fetch_add(n, std::memory_order_release);
  return Status::Ok();
}

// the upper layers of a point whose level 0 the device links (hnsw_build.hip): insert() without layer 0
Status HnswGraph::bulk_link_upper(uint32_t id) {
  const int node_top = levels_[id];
  if (node_top <= 0) return Status::Ok();
  const float *q = row(id);
  std::unique_lock<std::mutex> raising(global_);
  const int top = maxlevel_;
  Spin mine(link_locks_[id]);
  if (node_top <= top) raising.unlock();
  const uint32_t entry = enterpoint_;
  if (entry == kNone) return Status::Err(kErrInternal, "bulk insert into an empty graph");
  const uint32_t start = node_top < top ? descend(q, entry, top, node_top) : entry;
  VK_TRY(link_layers(id, q, start, std::min(node_top, top), 1, entry, /*rewire=*/false));
  if (node_top > top) {
    enterpoint_ = id;
    maxlevel_ = node_top;
  }
  return Status::Ok();
}

// ---- load path ----------------------------------------------------------------------------------
Status HnswGraph::load_element(uint32_t id, const uint32_t *links0_words, const float *new_row, uint64_t label) {
  if (id >= max_elements_) return Status::Err(kErrInternal, "element id beyond max_elements");
  VK_TRY(ensure_row_chunk(id));
  memcpy(links0_mut(id), links0_words, (maxM0_ + 1) * sizeof(uint32_t));
  memcpy(row_mut(id), new_row, dim_ * sizeof(float));
  labels_[id] = label;
  levels_[id] = 0;
  if (links0(id)[0] & kDeleteFlag) {
    num_deleted_ += 1;
    if (allow_replace_deleted_) vacant_.insert(id);
  }
  mark(id, 0);
  return Status::Ok();
}

// A label may sit on one live slot and any number of tombstoned ones in snapshots written
// by older versions; the lookup must point at the live slot (:1033-1052).
Status HnswGraph::load_labels(size_t count) {
  for (uint32_t i = 0; i < count; ++i) {
    if (labels_[i] == ~0ull && !is_deleted(i))
      return Status::Err(1 /* VK_ERR_INVALID */, "label UINT64_MAX is reserved (the padding of result lists)");
    auto it = label_lookup_.find(labels_[i]);
    if (it == label_lookup_.end()) {
      label_lookup_[labels_[i]] = i;
      note_label(labels_[i]);
    } else if (!is_deleted(i)) {
      if (!is_deleted(it->second))
        return Status::Err(kErrInternal, "HNSW index load validation failed: duplicate live label in index");
      it->second = i;
    }
  }
  return Status::Ok();
}

Status HnswGraph::load_upper(uint32_t id, const uint32_t *words, size_t n_words) {
  if (n_words == 0) { levels_[id] = 0; return Status::Ok(); }
  if (n_words % (maxM_ + 1)) return Status::Err(kErrInternal, "upper link list size is not a multiple of the per-level size");
  const int lv = (int)(n_words / (maxM_ + 1));
  levels_[id] = lv;
  delete[] upper_[id];
  upper_[id] = new uint32_t[n_words];
  memcpy(upper_[id], words, n_words * sizeof(uint32_t));
  upper_slot_[id] = upper_slots_used_.fetch_add((uint32_t)lv);
  mark(id, 1);
  return Status::Ok();
}

void HnswGraph::load_finish(size_t count, int maxlevel, uint32_t enterpoint) {
  count_.store(count);
  maxlevel_ = maxlevel;
  enterpoint_ = enterpoint;
}

}  // namespace vk
