// dispatcher.hpp -- N1: single-query searches -> device batches, several batches in flight.
//
// valkey-search has no batch API: every FT.SEARCH is one query vector.  query::SearchAsync puts it on the reader pool
// (src/query/search.cc:886-910; up to max-query-queue-depth = 100 000 requests wait there,
// src/valkey_search_options.cc:231-234) and a pool thread calls VectorFlat/VectorHNSW::Search for it (:135-170).  One
// query cannot feed an MI355X, so single-query requests that agree on (k, ef) -- one LANE -- are merged into one
// vk_index_search_batch, each with its own filter bitmap (or none) and its own cancellation token.
//
// Two ways in:
//   submit()  non-blocking (vk_index_search_submit): the request is queued and the call returns; a completion callback
//             fires from a dispatcher thread.  Queries in flight are bounded by max-query-queue-depth, not by the
//             number of reader threads -- the shape of SearchAsync.
//   search()  blocking (vk_index_search with coalescing on): submit + wait on the request's own futex word.
//
// `batches-in-flight` RUNNER threads (default 2) form and run the batches: while batch N is on the device, the next
// runner collects batch N+1, uploads it and enqueues its kernels behind (every batch has its own search context and
// stream), so the device does not idle while N's callers are woken and their answers copied out (r03: one leader, one
// batch in flight per lane -- 55-75 % of the device rate for FLAT, 20 % for HNSW).  A runner that finds less than a full
// batch waits until the batch is full, the oldest request has waited max_wait_us, or nobody has arrived for a quarter
// of that (20-200 us): callers of the batch that just finished come back within microseconds of each other.
//
// Cancellation.  A request whose token is up when its batch is formed is answered at once without a search
// (VK_ERR_CANCELLED for HNSW without partial results, vector_hnsw.cc:327-329; else an empty answer).  A BLOCKING caller
// whose token goes up while it is still queued leaves at once.  Once a request is in a batch the runner reads the
// caller's query and bitmap without the lock, so the caller stays until the batch returns; the batch carries ITS OWN
// cancellation word (SearchRequest::cancel_flag), raised by the watcher thread when every member of the batch has a
// token and all of them are up -- the kernels then stop within a millisecond (the reference polls the token inside the
// search: hnswalg.h:400-402, bruteforce.h:129).  Worst case for one cancelled member among live ones: the rest of its
// batch (a few milliseconds), documented in INTEGRATION.md.
#pragma once
#include <linux/futex.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "index.hpp"

namespace vk {

typedef void (*SearchDoneFn)(void *user, int status);   // == vk_search_done_fn (include/vk_index.h)

class Dispatcher {
 public:
  explicit Dispatcher(Index *ix) : ix_(ix) {}
  ~Dispatcher() { shutdown(); }
  Dispatcher(const Dispatcher &) = delete;
  Dispatcher &operator=(const Dispatcher &) = delete;

  void configure(uint32_t max_batch, uint32_t max_wait_us) {
    std::lock_guard<std::mutex> lk(mu_);
    max_batch_.store(max_batch, std::memory_order_relaxed);
    max_wait_us_ = max_wait_us;
    cv_.notify_all();   // a runner waiting for a batch that can no longer fill re-reads the limit
  }
  void set_in_flight(uint32_t n) {
    std::lock_guard<std::mutex> lk(mu_);
    in_flight_ = std::max<uint32_t>(1, std::min<uint32_t>(n, 8));
    cv_.notify_all();
  }
  void set_queue_depth(uint64_t d) { queue_depth_.store(d, std::memory_order_relaxed); }
  bool enabled() const { return max_batch_.load(std::memory_order_relaxed) > 1; }
  uint64_t batches() const { return batches_.load(std::memory_order_relaxed); }
  uint64_t queries() const { return queries_.load(std::memory_order_relaxed); }
  uint64_t submitted() const { return submitted_.load(std::memory_order_relaxed); }
  uint64_t rejected() const { return rejected_.load(std::memory_order_relaxed); }
  uint64_t queued() const { return queued_.load(std::memory_order_relaxed); }
  uint64_t max_in_flight_seen() const { return max_active_seen_.load(std::memory_order_relaxed); }

  // vk_index_search_submit: queue one single-query request and return.  `done(user, status)` is called exactly once, from
  // a dispatcher thread, after the outputs have been written (status = the vk_status of the batch the request travelled
  // in, or of the request alone).  Query, bitmap, token and output buffers must stay valid until then.
  Status submit(const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                const volatile int *cancel_flag, bool partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n,
                SearchDoneFn done, void *user) {
    auto r = std::make_shared<Req>();
    fill(*r, query, k, ef, allow_bits, allow_nbits, cancel_flag, partial_ok, out_dist, out_label, out_n);
    r->cb = done;
    r->user = user;
    return enqueue(r, /*bounded=*/true);
  }

  // vk_index_search with coalescing on: the same queue, the caller waits for its answer.
  Status search(const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                const volatile int *cancel_flag, bool partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
    if (cancel_raised(cancel_flag)) return cancelled_answer(partial_ok, out_n);
    auto r = std::make_shared<Req>();
    fill(*r, query, k, ef, allow_bits, allow_nbits, cancel_flag, partial_ok, out_dist, out_label, out_n);
    VK_TRY(enqueue(r, /*bounded=*/false));
    for (;;) {
      const uint32_t s = r->state.load(std::memory_order_acquire);
      if (s == kDone) break;
      if (s == kQueued && cancel_raised(cancel_flag)) {
        // leave -- but only while the request is still QUEUED: once a runner has popped it (state changes under mu_) the
        // runner reads the caller's query and bitmap with the lock released, and the module frees both when this returns
        std::unique_lock<std::mutex> lk(mu_);
        if (r->state.load(std::memory_order_relaxed) == kQueued) {
          auto it = lanes_.find(std::make_pair(k, ef));
          if (it != lanes_.end()) {
            auto &q = it->second.q;
            for (auto qi = q.begin(); qi != q.end(); ++qi)
              if (qi->get() == r.get()) { q.erase(qi); queued_.fetch_sub(1, std::memory_order_relaxed); break; }
            if (q.empty() && !it->second.collector) lanes_.erase(it);
          }
          r->state.store(kAbandoned, std::memory_order_relaxed);
          lk.unlock();
          return cancelled_answer(partial_ok, out_n);
        }
        continue;
      }
      futex_wait(&r->state, s, cancel_flag && s == kQueued ? 100 : 2000);
    }
    return r->st;
  }

  // no new requests; what is queued is answered; runners and watcher joined.  (vk_index_destroy: callers must not submit
  // concurrently with the destruction of the index -- the reference drains its pools before an index goes away.)
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_) return;
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : runners_) t.join();
    runners_.clear();
    {
      std::lock_guard<std::mutex> lk(wmu_);
      stop_watch_.store(true, std::memory_order_relaxed);
    }
    wcv_.notify_all();
    if (watcher_.joinable()) watcher_.join();
  }

 private:
  enum : uint32_t { kQueued = 0, kInBatch = 1, kDone = 2, kAbandoned = 3 };
  struct Req {
    const float *q = nullptr;
    const uint64_t *allow = nullptr;
    uint64_t allow_nbits = 0, k = 0, ef = 0;
    const volatile int *cancel = nullptr;
    bool partial_ok = true;
    float *od = nullptr;
    uint64_t *ol = nullptr;
    uint64_t *on = nullptr;
    SearchDoneFn cb = nullptr;
    void *user = nullptr;
    Status st;
    std::atomic<uint32_t> state{kQueued};
    std::chrono::steady_clock::time_point t_submit{};
  };
  struct Lane {
    std::deque<std::shared_ptr<Req>> q;
    bool collector = false;        // a runner is forming a batch from this lane
    std::chrono::steady_clock::time_point last_arrival{};
  };
  typedef std::pair<uint64_t, uint64_t> Key;
  // a batch on the device whose members all carry tokens: the watcher raises `word` when all of them are up
  struct Watched {
    std::vector<const volatile int *> flags;
    int word = 0;
  };

  static void fill(Req &r, const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                   const volatile int *cancel_flag, bool partial_ok, float *od, uint64_t *ol, uint64_t *on) {
    r.q = query; r.k = k; r.ef = ef; r.allow = allow_bits; r.allow_nbits = allow_nbits; r.cancel = cancel_flag;
    r.partial_ok = partial_ok; r.od = od; r.ol = ol; r.on = on;
  }
  Status cancelled_answer(bool partial_ok, uint64_t *out_n) const {
    *out_n = 0;
    const bool hnsw = ix_->params().algo == VK_ALGO_HNSW;
    return hnsw && !partial_ok ? Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout") : Status::Ok();
  }
  static void futex_wait(std::atomic<uint32_t> *w, uint32_t expect, long timeout_us) {
    struct timespec ts = {timeout_us / 1000000, (timeout_us % 1000000) * 1000};
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, expect, &ts, nullptr, 0);
  }
  static void futex_wake(std::atomic<uint32_t> *w) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
  }
  size_t batch_cap() const { return std::max<uint32_t>(1u, max_batch_.load(std::memory_order_relaxed)); }

  Status enqueue(const std::shared_ptr<Req> &r, bool bounded) {
    r->t_submit = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(mu_);
    if (stop_) return Status::Err(VK_ERR_INVALID, "the index is being destroyed");
    const uint64_t depth = queue_depth_.load(std::memory_order_relaxed);
    if (bounded && depth != 0 && queued_.load(std::memory_order_relaxed) >= depth) {
      rejected_.fetch_add(1, std::memory_order_relaxed);
      return Status::Err(VK_ERR_BUSY, "query queue is full (max-query-queue-depth)");
    }
    Lane &lane = lanes_[std::make_pair(r->k, r->ef)];
    lane.q.push_back(r);
    lane.last_arrival = r->t_submit;
    const uint64_t nq = queued_.fetch_add(1, std::memory_order_relaxed) + 1;
    submitted_.fetch_add(1, std::memory_order_relaxed);
    while (runners_.size() < in_flight_) runners_.emplace_back([this] { run(); });
    // wake a runner when there is one with nothing to do, or when the batch being collected is full
    const bool full = lane.collector && lane.q.size() >= batch_cap();
    if (idle_runners_ > 0 || full || nq == 1) cv_.notify_all();
    return Status::Ok();
  }

  // a lane that has requests and nobody collecting from it: the one whose head has waited longest
  std::map<Key, Lane>::iterator pick_lane() {
    auto best = lanes_.end();
    for (auto it = lanes_.begin(); it != lanes_.end(); ++it) {
      if (it->second.q.empty() || it->second.collector) continue;
      if (best == lanes_.end() || it->second.q.front()->t_submit < best->second.q.front()->t_submit) best = it;
    }
    return best;
  }

  void run() {
    std::vector<std::shared_ptr<Req>> batch;
    std::vector<float> D;
    std::vector<uint64_t> L, N, nbits;
    std::vector<const float *> qtab;
    std::vector<const uint64_t *> atab;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      auto it = lanes_.end();
      idle_runners_ += 1;
      cv_.wait(lk, [&] {
        if (active_ < in_flight_) it = pick_lane();
        return it != lanes_.end() || (stop_ && queued_.load(std::memory_order_relaxed) == 0);
      });
      idle_runners_ -= 1;
      if (it == lanes_.end()) return;   // stopping and drained
      const Key key = it->first;
      Lane *lane = &it->second;
      lane->collector = true;
      // the batching window (see the header)
      const auto quiet = std::chrono::microseconds(std::min<uint32_t>(200, std::max<uint32_t>(20, max_wait_us_ / 4)));
      while (!stop_ && !lane->q.empty() && lane->q.size() < batch_cap()) {
        const auto now = std::chrono::steady_clock::now();
        const auto deadline = lane->q.front()->t_submit + std::chrono::microseconds(max_wait_us_);
        if (now >= deadline) break;
        const auto since = now - lane->last_arrival;
        if (since >= quiet) break;
        cv_.wait_for(lk, std::min<std::chrono::steady_clock::duration>(deadline - now, quiet - since));
      }
      // (the limit is re-read: coalescing switched off while requests are queued must still drain them)
      const size_t cap = batch_cap();
      batch.clear();
      while (!lane->q.empty() && batch.size() < cap) {
        lane->q.front()->state.store(kInBatch, std::memory_order_relaxed);
        batch.push_back(std::move(lane->q.front()));
        lane->q.pop_front();
      }
      queued_.fetch_sub(batch.size(), std::memory_order_relaxed);
      lane->collector = false;
      if (lane->q.empty()) lanes_.erase(key);   // the map does not grow by one entry per (k, ef) pair ever seen
      else cv_.notify_all();                    // more than a batch was queued: another runner may start on the rest
      if (batch.empty()) continue;              // (every queued request of the lane left, cancelled)
      active_ += 1;
      if (active_ > max_active_seen_.load(std::memory_order_relaxed)) max_active_seen_.store(active_, std::memory_order_relaxed);
      lk.unlock();
      run_batch(key.first, key.second, batch, D, L, N, nbits, qtab, atab);
      batch.clear();
      lk.lock();
      active_ -= 1;
      cv_.notify_all();
    }
  }

  void complete(Req &r, const Status &st) {
    if (r.cb) {
      r.cb(r.user, st.code);
    } else {
      r.st = st;
      r.state.store(kDone, std::memory_order_release);
      futex_wake(&r.state);
    }
  }

  void run_batch(uint64_t k, uint64_t ef, std::vector<std::shared_ptr<Req>> &batch, std::vector<float> &D, std::vector<uint64_t> &L,
                 std::vector<uint64_t> &N, std::vector<uint64_t> &nbits, std::vector<const float *> &qtab,
                 std::vector<const uint64_t *> &atab) {
    const bool hnsw = ix_->params().algo == VK_ALGO_HNSW;
    // requests whose token is already up are answered without a search
    size_t live = 0;
    for (size_t i = 0; i < batch.size(); ++i) {
      Req &r = *batch[i];
      if (cancel_raised(r.cancel)) {
        *r.on = 0;
        complete(r, hnsw && !r.partial_ok ? Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout") : Status::Ok());
      } else {
        if (live != i) batch[live] = std::move(batch[i]);
        ++live;
      }
    }
    batch.resize(live);
    const uint64_t nq = batch.size();
    if (nq == 0) return;
    Status st = Status::Ok();
    std::shared_ptr<Watched> w;
    try {
      D.resize(nq * k);
      L.resize(nq * k);
      N.assign(nq, 0);
      qtab.resize(nq);
      bool any_filter = false, all_tokens = true;
      for (uint64_t i = 0; i < nq; ++i) {
        qtab[i] = batch[i]->q;
        any_filter = any_filter || batch[i]->allow != nullptr;
        all_tokens = all_tokens && batch[i]->cancel != nullptr;
      }
      SearchRequest rq;
      rq.query_tab = qtab.data();
      rq.nq = nq;
      rq.k = k;
      rq.ef = ef;
      rq.partial_ok = true;   // (per-member rule below)
      if (any_filter) {
        atab.resize(nq);
        nbits.resize(nq);
        for (uint64_t i = 0; i < nq; ++i) { atab[i] = batch[i]->allow; nbits[i] = batch[i]->allow_nbits; }
        rq.allow_tab = atab.data();
        rq.allow_nbits_tab = nbits.data();
      }
      if (all_tokens) {   // the batch's own cancellation word, raised by the watcher when every member's token is up
        w = std::make_shared<Watched>();
        w->flags.reserve(nq);
        for (uint64_t i = 0; i < nq; ++i) w->flags.push_back(batch[i]->cancel);
        rq.cancel_flag = &w->word;
        watch(w);
      }
      st = ix_->search(rq, D.data(), L.data(), N.data());
    } catch (const std::exception &e) {
      st = Status::Err(VK_ERR_INTERNAL, e.what());
    }
    if (w) unwatch(w);
    batches_.fetch_add(1, std::memory_order_relaxed);
    queries_.fetch_add(nq, std::memory_order_relaxed);
    for (uint64_t i = 0; i < nq; ++i) {
      Req &r = *batch[i];
      Status mine = st;
      if (st.ok()) {
        *r.on = N[i];
        if (N[i]) {
          memcpy(r.od, D.data() + i * k, (size_t)N[i] * 4);
          memcpy(r.ol, L.data() + i * k, (size_t)N[i] * 8);
        }
        if (hnsw && !r.partial_ok && cancel_raised(r.cancel)) {
          // cancelled while the batch it travelled in was on the device: the reference's answer for a raised token
          // (vector_hnsw.cc:327-329), whatever the batch found
          *r.on = 0;
          mine = Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
        }
      }
      complete(r, mine);
    }
  }

  // ---- the watcher: batches on the device whose members all have tokens --------------------------------------------
  void watch(const std::shared_ptr<Watched> &w) {
    std::lock_guard<std::mutex> lk(wmu_);
    watched_.push_back(w);
    if (!watcher_.joinable()) watcher_ = std::thread([this] { watch_loop(); });
    wcv_.notify_one();
  }
  void unwatch(const std::shared_ptr<Watched> &w) {
    std::lock_guard<std::mutex> lk(wmu_);
    watched_.erase(std::remove(watched_.begin(), watched_.end(), w), watched_.end());
  }
  void watch_loop() {
    std::unique_lock<std::mutex> lk(wmu_);
    for (;;) {
      if (watched_.empty()) wcv_.wait(lk, [&] { return !watched_.empty() || stop_flag(); });
      else wcv_.wait_for(lk, std::chrono::microseconds(200));
      if (stop_flag() && watched_.empty()) return;
      for (auto &w : watched_) {
        if (__atomic_load_n(&w->word, __ATOMIC_RELAXED)) continue;
        bool all = true;
        for (const volatile int *f : w->flags)
          if (!cancel_raised(f)) { all = false; break; }
        if (all) __atomic_store_n(&w->word, 1, __ATOMIC_RELAXED);
      }
    }
  }
  bool stop_flag() const { return stop_watch_.load(std::memory_order_relaxed); }

  Index *ix_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::map<Key, Lane> lanes_;
  std::vector<std::thread> runners_;
  uint32_t in_flight_ = 2, active_ = 0, idle_runners_ = 0;
  bool stop_ = false;
  std::atomic<uint32_t> max_batch_{0};
  uint32_t max_wait_us_ = 0;
  std::atomic<uint64_t> queue_depth_{100000};
  std::atomic<uint64_t> queued_{0}, batches_{0}, queries_{0}, submitted_{0}, rejected_{0}, max_active_seen_{0};
  std::mutex wmu_;
  std::condition_variable wcv_;
  std::vector<std::shared_ptr<Watched>> watched_;
  std::thread watcher_;
  std::atomic<bool> stop_watch_{false};
};

}  // namespace vk
