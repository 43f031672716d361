// coalescer.hpp -- N1: query coalescing between the reader pool and the device.
//
// valkey-search has no batch API: every FT.SEARCH is one query vector and one call to
// VectorFlat/VectorHNSW::Search on a reader-pool thread (src/query/search.cc:135-170, :886-910),
// up to `reader-threads` of them concurrently.  One query cannot feed an MI355X (a FLAT scan is
// HBM-bound at any batch size up to ~dozens, the MFMA path wants >= 16 queries), so concurrent
// single-query calls that are compatible (same k, same ef, no filter, no cancel flag) are merged
// into one vk_index_search_batch: the first caller to arrive becomes the leader, waits until
// `max_batch` requests are queued, `max_wait_us` elapsed or arrivals have stopped, runs the batch,
// and hands every follower its slice.  Latency/throughput knob, off by default (max_batch <= 1).
#pragma once
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "index.hpp"

namespace vk {

class Coalescer {
 public:
  void configure(uint32_t max_batch, uint32_t max_wait_us) {
    std::lock_guard<std::mutex> lk(mu_);
    max_batch_.store(max_batch, std::memory_order_relaxed);
    max_wait_us_ = max_wait_us;
    for (auto &kv : lanes_)   // a leader waiting for a batch that can no longer fill re-reads the limit
      if (kv.second.leader) kv.second.leader->cv.notify_one();
  }
  bool enabled() const { return max_batch_.load(std::memory_order_relaxed) > 1; }
  uint64_t batches() const { return batches_; }
  uint64_t queries() const { return queries_; }

  // One single-query request; returns the status of the batch it travelled in.  Requests of one (k, ef) lane travel
  // together whatever their filters: each carries its own allow-bitmap (or none) and the batch is searched with one
  // filter per query (SearchRequest::allow_tab) -- hybrid FT.SEARCH traffic batches like plain traffic.  A request
  // with a cancellation flag keeps watching it while it waits: once raised, the caller leaves at once (HNSW without
  // partial results: VK_ERR_CANCELLED, vector_hnsw.cc:327-329; otherwise an empty answer, "what it has") and the
  // batch's answer for it is dropped.
  Status search(Index *ix, const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                const volatile int *cancel_flag, bool partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
    auto me = std::make_shared<Req>();
    me->q = query;
    me->allow = allow_bits;
    me->allow_nbits = allow_nbits;
    me->od = out_dist;
    me->ol = out_label;
    me->on = out_n;
    std::unique_lock<std::mutex> lk(mu_);
    const auto key = std::make_pair(k, ef);
    Lane &lane = lanes_[key];
    lane.q.push_back(me);
    lane.last_arrival = std::chrono::steady_clock::now();
    // Every request waits on its own condition variable and is woken by name: the leader when its batch is full, the
    // members of a batch when their answers are in, the request at the head of the queue when the lane needs a new
    // leader.  (One shared variable and notify_all woke every waiting caller at every batch: with 1024 callers on 16
    // CPUs a batch took 36 ms of wake-ups and lock hand-overs around 3 ms of device time.)
    if (lane.leader && lane.q.size() >= batch_cap()) lane.leader->cv.notify_one();  // batch full: wake the leader
    while (!me->done) {
      // A raised token lets the caller leave only while its request is still QUEUED.  Once a leader has popped it into
      // a batch (in_batch, set under this mutex) the leader reads the caller's query and allow-bitmap with the mutex
      // released -- for tens of milliseconds on a filtered batch -- and the module frees both as soon as this call
      // returns, so from then on the caller waits for `done` and only its answer is dropped.
      if (cancel_raised(cancel_flag) && !me->in_batch) {
        me->abandoned = true;
        for (auto it = lane.q.begin(); it != lane.q.end(); ++it)
          if (it->get() == me.get()) { lane.q.erase(it); break; }
        // (this request may have been the one woken to drive the lane: pass that on)
        if (!lane.leader_active && !lane.q.empty()) lane.q.front()->cv.notify_one();
        *out_n = 0;
        const bool hnsw = ix->params().algo == VK_ALGO_HNSW;
        return hnsw && !partial_ok ? Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout") : Status::Ok();
      }
      if (lane.leader_active) {
        if (cancel_flag) me->cv.wait_for(lk, std::chrono::microseconds(100));
        else me->cv.wait(lk);
        continue;
      }
      // nobody is driving this lane: lead one batch (ours is in it unless the queue is longer
      // than max_batch, in which case the loop leads or follows again)
      lane.leader_active = true;
      lane.leader = me.get();
      lead_one_batch(ix, lane, k, ef, lk, cancel_flag, me.get());
      lane.leader_active = false;
      lane.leader = nullptr;
      if (!lane.q.empty()) lane.q.front()->cv.notify_one();   // whoever waits longest drives the next batch
    }
    // (`lane` may be gone by now: it is not touched after `done`.)  Drop the lane of this (k, ef) once it is idle,
    // so the map does not grow by one entry per distinct pair ever seen
    auto it = lanes_.find(key);
    if (it != lanes_.end() && it->second.q.empty() && !it->second.leader_active) lanes_.erase(it);
    if (me->st.ok() && cancel_raised(cancel_flag) && !partial_ok && ix->params().algo == VK_ALGO_HNSW) {
      // cancelled while the batch it travelled in was on the device: the reference's answer for a raised token
      // (vector_hnsw.cc:327-329), whatever the batch found
      *out_n = 0;
      return Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
    }
    return me->st;
  }

 private:
  struct Req {
    const float *q = nullptr;
    const uint64_t *allow = nullptr;
    uint64_t allow_nbits = 0;
    float *od = nullptr;
    uint64_t *ol = nullptr;
    uint64_t *on = nullptr;
    Status st;
    bool done = false, abandoned = false;
    bool in_batch = false;   // popped by a leader: its input pointers are being read without the mutex
    std::condition_variable cv;
  };
  struct Lane {
    std::deque<std::shared_ptr<Req>> q;
    bool leader_active = false;
    Req *leader = nullptr;      // valid while leader_active (the request lives on its caller's stack frame via `me`)
    std::chrono::steady_clock::time_point last_arrival{};
  };
  void lead_one_batch(Index *ix, Lane &lane, uint64_t k, uint64_t ef, std::unique_lock<std::mutex> &lk,
                      const volatile int *leader_cancel, Req *self) {
    const uint32_t dim = ix->params().dim;
    const auto start = std::chrono::steady_clock::now();
    const auto deadline = start + std::chrono::microseconds(max_wait_us_);
    // The leader waits for company until the batch is full, max_wait_us has passed -- or nobody has arrived for a
    // quarter of it (20-200 us): the callers of the batch that just finished come back within microseconds of each
    // other, and once they are in, waiting out the rest of max_wait_us only idles the device.
    // (a leader whose own token is raised stops waiting for company and runs what is queued)
    const auto quiet = std::chrono::microseconds(std::min<uint32_t>(200, std::max<uint32_t>(20, max_wait_us_ / 4)));
    while (lane.q.size() < batch_cap() && !cancel_raised(leader_cancel)) {
      const auto now = std::chrono::steady_clock::now();
      if (now >= deadline) break;
      const auto since = now - std::max(lane.last_arrival, start);
      if (since >= quiet) break;
      auto slice = std::min<std::chrono::steady_clock::duration>(deadline - now, quiet - since);
      if (leader_cancel) slice = std::min<std::chrono::steady_clock::duration>(slice, std::chrono::microseconds(100));
      self->cv.wait_for(lk, slice);
    }
    // (the limit is re-read: vk_index_set_coalescing(ix, 0, ..) while requests are queued must still drain them --
    // a leader always takes at least its own request)
    const size_t cap = batch_cap();
    std::vector<std::shared_ptr<Req>> batch;
    while (!lane.q.empty() && batch.size() < cap) {
      lane.q.front()->in_batch = true;
      batch.push_back(lane.q.front());
      lane.q.pop_front();
    }
    lk.unlock();
    const uint64_t nq = batch.size();
    Status st = Status::Ok();
    std::vector<float> Q, D;
    std::vector<uint64_t> L, N, nbits;
    std::vector<const uint64_t *> tab;
    try {
      Q.resize(nq * dim);
      D.resize(nq * k);
      L.resize(nq * k);
      N.resize(nq);
      bool any_filter = false;
      for (uint64_t i = 0; i < nq; ++i) {
        memcpy(Q.data() + i * dim, batch[i]->q, (size_t)dim * 4);
        any_filter = any_filter || batch[i]->allow != nullptr;
      }
      SearchRequest rq;
      rq.queries = Q.data();
      rq.nq = nq;
      rq.k = k;
      rq.ef = ef;
      if (any_filter) {
        tab.resize(nq);
        nbits.resize(nq);
        for (uint64_t i = 0; i < nq; ++i) { tab[i] = batch[i]->allow; nbits[i] = batch[i]->allow_nbits; }
        rq.allow_tab = tab.data();
        rq.allow_nbits_tab = nbits.data();
      }
      if (nq) st = ix->search(rq, D.data(), L.data(), N.data());
    } catch (const std::exception &e) {
      st = Status::Err(VK_ERR_INTERNAL, e.what());
    }
    lk.lock();
    batches_ += 1;
    queries_ += nq;
    for (uint64_t i = 0; i < nq; ++i) {
      Req *r = batch[i].get();
      if (r->abandoned) continue;          // its caller has left (cancelled): the buffers are no longer ours to write
      r->st = st;
      if (st.ok()) {
        *r->on = N[i];
        if (N[i]) {
          memcpy(r->od, D.data() + i * k, (size_t)N[i] * 4);
          memcpy(r->ol, L.data() + i * k, (size_t)N[i] * 8);
        }
      }
      r->done = true;
      if (r != self) r->cv.notify_one();
    }
  }

  size_t batch_cap() const { return std::max<uint32_t>(1u, max_batch_.load(std::memory_order_relaxed)); }

  std::mutex mu_;
  std::map<std::pair<uint64_t, uint64_t>, Lane> lanes_;
  std::atomic<uint32_t> max_batch_{0};
  uint32_t max_wait_us_ = 0;
  uint64_t batches_ = 0, queries_ = 0;
};

}  // namespace vk
