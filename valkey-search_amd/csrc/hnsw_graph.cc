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
  level_generator_.seed(seed);                       // :149
  update_probability_generator_.seed(seed + 1);
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

std::unique_ptr<HnswGraph::VisitedList> HnswGraph::get_visited() {
  {
    std::lock_guard<std::mutex> lk(visited_mu_);
    if (!visited_pool_.empty()) {
      auto v = std::move(visited_pool_.back());
      visited_pool_.pop_back();
      if (v->mass.size() >= max_elements_) return v;
    }
  }
  return std::make_unique<VisitedList>(max_elements_ ? max_elements_ : 1);
}
void HnswGraph::put_visited(std::unique_ptr<VisitedList> v) {
  std::lock_guard<std::mutex> lk(visited_mu_);
  visited_pool_.push_back(std::move(v));
}

int HnswGraph::random_level() {                        // :243-247
  std::uniform_real_distribution<double> distribution(0.0, 1.0);
  double r = -log(distribution(level_generator_)) * mult_;
  return (int)r;
}

Status HnswGraph::resize(size_t new_max) {             // :758-777
  if (new_max < count_.load())
    return Status::Err(kErrInternal, "Cannot resize, max element is less than the current number of elements");
  alloc_tables(new_max, std::min(count_.load(), max_elements_));
  max_elements_ = new_max;
  std::lock_guard<std::mutex> lk(visited_mu_);
  visited_pool_.clear();
  return Status::Ok();
}

// ---- :255-347 --------------------------------------------------------------------------------
HnswGraph::Heap HnswGraph::search_base_layer(uint32_t ep_id, const float *q, int layer) {
  auto vl = get_visited();
  uint16_t *visited = vl->mass.data();
  const uint16_t tag = vl->next();
  Heap top, cand;
  float lowerBound;
  if (!is_deleted(ep_id)) {
    float d = dist(q, row(ep_id));
    top.emplace(d, ep_id);
    lowerBound = d;
    cand.emplace(-d, ep_id);
  } else {
    lowerBound = std::numeric_limits<float>::max();
    cand.emplace(-lowerBound, ep_id);
  }
  visited[ep_id] = tag;
  std::vector<uint32_t> nbrs;
  nbrs.reserve(maxM0_);
  while (!cand.empty()) {
    Pair cur = cand.top();
    if ((-cur.first) > lowerBound && top.size() == efC_) break;
    cand.pop();
    const uint32_t cur_id = cur.second;
    {
      Spin lock(link_locks_[cur_id]);
      const uint32_t *ll = list_at(cur_id, layer);
      size_t size = list_count(ll);
      nbrs.assign(ll + 1, ll + 1 + size);
    }
    for (uint32_t cid : nbrs) {
      if (visited[cid] == tag) continue;
      visited[cid] = tag;
      float d1 = dist(q, row(cid));
      if (top.size() < efC_ || lowerBound > d1) {
        cand.emplace(-d1, cid);
        if (!is_deleted(cid)) top.emplace(d1, cid);
        if (top.size() > efC_) top.pop();
        if (!top.empty()) lowerBound = top.top().first;
      }
    }
  }
  put_visited(std::move(vl));
  return top;
}

// ---- :553-594 --------------------------------------------------------------------------------
void HnswGraph::neighbors_by_heuristic2(Heap &top, size_t M) {
  if (top.size() < M) return;
  std::priority_queue<Pair> queue_closest;   // std::less<pair>: ties on distance order by id
  std::vector<Pair> return_list;
  while (!top.empty()) {
    queue_closest.emplace(-top.top().first, top.top().second);
    top.pop();
  }
  while (!queue_closest.empty()) {
    if (return_list.size() >= M) break;
    Pair cur = queue_closest.top();
    float dist_to_query = -cur.first;
    queue_closest.pop();
    bool good = true;
    for (const Pair &second : return_list) {
      float curdist = dist(row(second.second), row(cur.second));
      if (curdist < dist_to_query) { good = false; break; }
    }
    if (good) return_list.push_back(cur);
  }
  for (const Pair &p : return_list) top.emplace(-p.first, p.second);
}

// ---- :613-756 --------------------------------------------------------------------------------
Status HnswGraph::mutually_connect(const float *q, uint32_t cur_c, Heap &top, int level, bool is_update, uint32_t *next) {
  (void)q;
  const size_t Mcurmax = level ? maxM_ : maxM0_;
  neighbors_by_heuristic2(top, M_);
  if (top.size() > M_) return Status::Err(kErrInternal, "Should be not be more than M_ candidates returned by the heuristic");
  std::vector<uint32_t> sel;
  sel.reserve(M_);
  while (!top.empty()) { sel.push_back(top.top().second); top.pop(); }
  if (sel.empty()) return Status::Err(kErrInternal, "During insertion, no neighbors found to mutually connect to");
  *next = sel.back();
  {
    // the lock for cur_c is already held during an insert (:646-650)
    std::unique_ptr<Spin> lock;
    if (is_update) lock = std::make_unique<Spin>(link_locks_[cur_c]);
    uint32_t *ll_cur = list_at(cur_c, level);
    if (*ll_cur && !is_update) return Status::Err(kErrInternal, "The newly inserted element should have blank link list");
    set_list_count(ll_cur, (unsigned)sel.size());
    for (size_t i = 0; i < sel.size(); ++i) {
      if (ll_cur[1 + i] && !is_update) return Status::Err(kErrInternal, "Possible memory corruption");
      if (level > levels_[sel[i]]) return Status::Err(kErrInternal, "Trying to make a link on a non-existent level");
      ll_cur[1 + i] = sel[i];
    }
    mark(cur_c, level);
  }
  for (size_t i = 0; i < sel.size(); ++i) {
    const uint32_t nb = sel[i];
    Spin lock(link_locks_[nb]);
    uint32_t *ll_other = list_at(nb, level);
    const size_t sz = list_count(ll_other);
    if (sz > Mcurmax) return Status::Err(kErrInternal, "Bad value of sz_link_list_other");
    if (nb == cur_c) return Status::Err(kErrInternal, "Trying to connect an element to itself");
    if (level > levels_[nb]) return Status::Err(kErrInternal, "Trying to make a link on a non-existent level");
    uint32_t *data = ll_other + 1;
    bool present = false;
    if (is_update)
      for (size_t j = 0; j < sz; ++j)
        if (data[j] == cur_c) { present = true; break; }
    if (present) continue;
    if (sz < Mcurmax) {
      data[sz] = cur_c;
      set_list_count(ll_other, (unsigned)(sz + 1));
    } else {
      float d_max = dist(row(cur_c), row(nb));
      Heap cands;
      cands.emplace(d_max, cur_c);
      for (size_t j = 0; j < sz; ++j) cands.emplace(dist(row(data[j]), row(nb)), data[j]);
      neighbors_by_heuristic2(cands, Mcurmax);
      unsigned indx = 0;
      while (!cands.empty()) { data[indx++] = cands.top().second; cands.pop(); }
      set_list_count(ll_other, indx);
    }
    mark(nb, level);
  }
  return Status::Ok();
}

Status HnswGraph::mark_deleted_internal(uint32_t id) {   // :1194-1209
  if (is_deleted(id)) return Status::Err(kErrInternal, "The requested to delete element is already deleted");
  __atomic_fetch_or(links0_mut(id), kDeleteFlag, __ATOMIC_RELAXED);   // (see set_list_count)
  num_deleted_ += 1;
  mark(id, 0);
  if (allow_replace_deleted_) {
    std::lock_guard<std::mutex> lk(deleted_lock_);
    deleted_elements_.insert(id);
  }
  return Status::Ok();
}

Status HnswGraph::unmark_deleted_internal(uint32_t id) { // :1236-1251
  if (!is_deleted(id)) return Status::Err(kErrInternal, "The requested to undelete element is not deleted");
  __atomic_fetch_and(links0_mut(id), ~kDeleteFlag, __ATOMIC_RELAXED);
  num_deleted_ -= 1;
  mark(id, 0);
  if (allow_replace_deleted_) {
    std::lock_guard<std::mutex> lk(deleted_lock_);
    deleted_elements_.erase(id);
  }
  return Status::Ok();
}

Status HnswGraph::mark_delete(uint64_t label) {          // :1173-1187
  std::lock_guard<std::mutex> lock_label(label_op_locks_[label & (kLabelLocks - 1)]);
  uint32_t id;
  if (!lookup(label, &id)) return Status::Err(kErrNotFound, "Label not found");
  return mark_deleted_internal(id);
}

std::vector<uint32_t> HnswGraph::connections_with_lock(uint32_t id, int level) {
  Spin lock(link_locks_[id]);
  const uint32_t *ll = list_at(id, level);
  return std::vector<uint32_t>(ll + 1, ll + 1 + list_count(ll));
}

// ---- :1342-1430 ------------------------------------------------------------------------------
Status HnswGraph::update_point(const float *new_row, uint32_t id, float prob) {
  memcpy(row_mut(id), new_row, dim_ * sizeof(float));
  const int maxLevelCopy = maxlevel_;
  const uint32_t entryPointCopy = enterpoint_;
  if (entryPointCopy == id && count_.load() == 1) return Status::Ok();
  const int elemLevel = levels_[id];
  std::uniform_real_distribution<float> distribution(0.0, 1.0);
  for (int layer = 0; layer <= elemLevel; layer++) {
    std::unordered_set<uint32_t> sCand, sNeigh;
    std::vector<uint32_t> listOneHop = connections_with_lock(id, layer);
    if (listOneHop.empty()) continue;
    sCand.insert(id);
    for (uint32_t elOneHop : listOneHop) {
      sCand.insert(elOneHop);
      float u;
      {
        std::lock_guard<std::mutex> lk(rng_mu_);
        u = distribution(update_probability_generator_);
      }
      if (u > prob) continue;
      sNeigh.insert(elOneHop);
      for (uint32_t elTwoHop : connections_with_lock(elOneHop, layer)) sCand.insert(elTwoHop);
    }
    for (uint32_t neigh : sNeigh) {
      Heap candidates;
      const size_t size = sCand.find(neigh) == sCand.end() ? sCand.size() : sCand.size() - 1;
      const size_t elementsToKeep = std::min(efC_, size);
      for (uint32_t cand : sCand) {
        if (cand == neigh) continue;
        float distance = dist(row(neigh), row(cand));
        if (candidates.size() < elementsToKeep) {
          candidates.emplace(distance, cand);
        } else if (!candidates.empty() && distance < candidates.top().first) {
          candidates.pop();
          candidates.emplace(distance, cand);
        }
      }
      neighbors_by_heuristic2(candidates, layer == 0 ? maxM0_ : maxM_);
      {
        Spin lock(link_locks_[neigh]);
        uint32_t *ll_cur = list_at(neigh, layer);
        const size_t candSize = candidates.size();
        set_list_count(ll_cur, (unsigned)candSize);
        for (size_t idx = 0; idx < candSize; idx++) { ll_cur[1 + idx] = candidates.top().second; candidates.pop(); }
        mark(neigh, layer);
      }
    }
  }
  return repair_connections(row(id), entryPointCopy, id, elemLevel, maxLevelCopy);
}

// ---- :1432-1511 ------------------------------------------------------------------------------
Status HnswGraph::repair_connections(const float *q, uint32_t ep, uint32_t id, int dataPointLevel, int maxLevel) {
  uint32_t currObj = ep;
  if (dataPointLevel < maxLevel) {
    float curdist = dist(q, row(currObj));
    for (int level = maxLevel; level > dataPointLevel; level--) {
      bool changed = true;
      while (changed) {
        changed = false;
        std::vector<uint32_t> nb = connections_with_lock(currObj, level);
        for (uint32_t cand : nb) {
          float d = dist(q, row(cand));
          if (d < curdist) { curdist = d; currObj = cand; changed = true; }
        }
      }
    }
  }
  if (dataPointLevel > maxLevel) return Status::Err(kErrInternal, "Level of item to be updated cannot be bigger than max level");
  for (int level = dataPointLevel; level >= 0; level--) {
    Heap topCandidates = search_base_layer(currObj, q, level);
    Heap filtered;
    while (!topCandidates.empty()) {
      if (topCandidates.top().second != id) filtered.push(topCandidates.top());
      topCandidates.pop();
    }
    if (!filtered.empty()) {
      if (is_deleted(ep)) {
        filtered.emplace(dist(q, row(ep)), ep);
        if (filtered.size() > efC_) filtered.pop();
      }
      VK_TRY(mutually_connect(q, id, filtered, level, true, &currObj));
    }
  }
  return Status::Ok();
}

// ---- :1523-1650 ------------------------------------------------------------------------------
Status HnswGraph::add_point_level(const float *new_row, uint64_t label, int level_in, uint32_t *out_id) {
  uint32_t cur_c = 0;
  {
    std::unique_lock<std::mutex> lock_table(label_lookup_lock_);
    auto search = label_lookup_.find(label);
    if (search != label_lookup_.end()) {
      const uint32_t existing = search->second;
      if (allow_replace_deleted_ && is_deleted(existing))
        return Status::Err(kErrInternal,
                           "Can't use addPoint to update deleted elements if replacement of deleted elements is enabled.");
      lock_table.unlock();
      if (is_deleted(existing)) VK_TRY(unmark_deleted_internal(existing));
      *out_id = existing;
      return update_point(new_row, existing, 1.0f);
    }
    if (count_.load() >= max_elements_)
      return Status::Err(kErrCapacity, "The number of elements exceeds the specified limit");
    cur_c = (uint32_t)count_.load();
    VK_TRY(ensure_row_chunk(cur_c));
    // initialise the slot before it becomes visible through count_ / label_lookup_
    memset(links0_mut(cur_c), 0, (maxM0_ + 1) * sizeof(uint32_t));
    labels_[cur_c] = label;
    memcpy(row_mut(cur_c), new_row, dim_ * sizeof(float));
    count_.fetch_add(1, std::memory_order_release);
    label_lookup_[label] = cur_c;
    note_label(label);
  }
  *out_id = cur_c;

  std::unique_lock<std::mutex> templock(global_);
  const int maxlevelcopy = maxlevel_;
  Spin lock_el(link_locks_[cur_c]);
  int curlevel = random_level();
  if (level_in > 0) curlevel = level_in;
  if (curlevel <= maxlevelcopy) templock.unlock();
  levels_[cur_c] = curlevel;
  uint32_t currObj = enterpoint_;
  const uint32_t enterpoint_copy = enterpoint_;

  delete[] upper_[cur_c];
  upper_[cur_c] = nullptr;
  if (curlevel) {
    upper_[cur_c] = new uint32_t[(size_t)curlevel * (maxM_ + 1)]();
    upper_slot_[cur_c] = upper_slots_used_.fetch_add((uint32_t)curlevel);
  }
  mark(cur_c, 0);
  if (curlevel) mark(cur_c, 1);

  if (currObj != kNone) {
    if (curlevel < maxlevelcopy) {
      float curdist = dist(new_row, row(currObj));
      for (int level = maxlevelcopy; level > curlevel; level--) {
        bool changed = true;
        while (changed) {
          changed = false;
          Spin lock(link_locks_[currObj]);
          const uint32_t *ll = upper(currObj, level);
          const int size = (int)list_count(ll);
          for (int i = 0; i < size; i++) {
            const uint32_t cand = ll[1 + i];
            if (cand > max_elements_) return Status::Err(kErrInternal, "cand error");
            float d = dist(new_row, row(cand));
            if (d < curdist) { curdist = d; currObj = cand; changed = true; }
          }
        }
      }
    }
    const bool epDeleted = is_deleted(enterpoint_copy);
    for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
      Heap top = search_base_layer(currObj, new_row, level);
      if (epDeleted) {
        top.emplace(dist(new_row, row(enterpoint_copy)), enterpoint_copy);
        if (top.size() > efC_) top.pop();
      }
      VK_TRY(mutually_connect(new_row, cur_c, top, level, false, &currObj));
    }
  } else {
    enterpoint_ = 0;
    maxlevel_ = curlevel;
  }
  if (curlevel > maxlevelcopy) {
    enterpoint_ = cur_c;
    maxlevel_ = curlevel;
  }
  return Status::Ok();
}

// ---- :1278-1340 ------------------------------------------------------------------------------
Status HnswGraph::add(const float *new_row, uint64_t label, uint32_t *out_id) {
  std::lock_guard<std::mutex> lock_label(label_op_locks_[label & (kLabelLocks - 1)]);
  if (!allow_replace_deleted_) return add_point_level(new_row, label, -1, out_id);
  {
    uint32_t existing;
    if (lookup(label, &existing)) {
      if (is_deleted(existing)) {
        {
          std::lock_guard<std::mutex> lk(deleted_lock_);
          deleted_elements_.erase(existing);
        }
        VK_TRY(unmark_deleted_internal(existing));
      }
      *out_id = existing;
      return update_point(new_row, existing, 1.0f);
    }
  }
  uint32_t replaced = 0;
  bool vacant = false;
  {
    std::lock_guard<std::mutex> lk(deleted_lock_);
    if (!deleted_elements_.empty()) {
      auto it = deleted_elements_.begin();
      replaced = *it;
      deleted_elements_.erase(it);
      vacant = true;
    }
  }
  if (!vacant) return add_point_level(new_row, label, -1, out_id);
  const uint64_t label_replaced = labels_[replaced];
  labels_[replaced] = label;
  {
    std::lock_guard<std::mutex> lk(label_lookup_lock_);
    label_lookup_.erase(label_replaced);
    label_lookup_[label] = replaced;
    note_label(label);
  }
  VK_TRY(unmark_deleted_internal(replaced));
  *out_id = replaced;
  return update_point(new_row, replaced, 1.0f);
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
    const int lv = random_level();
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

Status HnswGraph::bulk_link_upper(uint32_t id) {
  const int curlevel = levels_[id];
  if (curlevel <= 0) return Status::Ok();
  const float *q = row(id);
  std::unique_lock<std::mutex> templock(global_);
  const int maxlevelcopy = maxlevel_;
  Spin lock_el(link_locks_[id]);
  if (curlevel <= maxlevelcopy) templock.unlock();
  uint32_t currObj = enterpoint_;
  const uint32_t enterpoint_copy = enterpoint_;
  if (currObj == kNone) return Status::Err(kErrInternal, "bulk insert into an empty graph");
  if (curlevel < maxlevelcopy) {
    float curdist = dist(q, row(currObj));
    for (int level = maxlevelcopy; level > curlevel; level--) {
      bool changed = true;
      while (changed) {
        changed = false;
        Spin lock(link_locks_[currObj]);
        const uint32_t *ll = upper(currObj, level);
        const int size = (int)list_count(ll);
        for (int i = 0; i < size; i++) {
          const uint32_t cand = ll[1 + i];
          float d = dist(q, row(cand));
          if (d < curdist) { curdist = d; currObj = cand; changed = true; }
        }
      }
    }
  }
  const bool epDeleted = is_deleted(enterpoint_copy);
  for (int level = std::min(curlevel, maxlevelcopy); level >= 1; level--) {
    Heap top = search_base_layer(currObj, q, level);
    if (epDeleted) {
      top.emplace(dist(q, row(enterpoint_copy)), enterpoint_copy);
      if (top.size() > efC_) top.pop();
    }
    VK_TRY(mutually_connect(q, id, top, level, false, &currObj));
  }
  if (curlevel > maxlevelcopy) {
    enterpoint_ = id;
    maxlevel_ = curlevel;
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
    if (allow_replace_deleted_) deleted_elements_.insert(id);
  }
  mark(id, 0);
  return Status::Ok();
}

// A label may sit on one live slot and any number of tombstoned ones in snapshots written
// by older versions; the lookup must point at the live slot (:1033-1052).
Status HnswGraph::load_labels(size_t count) {
  for (uint32_t i = 0; i < count; ++i) {
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
