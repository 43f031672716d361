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
// COMPLETER threads (option completer-threads, default 6; the PROCESS's, shared by all its indexes: CompleterPool) hand the
// answers of a finished batch to the callers' callbacks, a piece of the batch each: a callback runs the caller's code (the adaptor builds the neighbour list and posts it on; 7-18 us
// each measured), and with the runner doing that a FLAT pass started 1.5 ms late (0.77 of the device rate through the
// adaptor, 0.98 with completers) and an HNSW runner spent more time answering than searching (0.35 -> 0.86).
// vk_index_stats.dispatch_*_us is the dispatcher's own account of where its threads' time goes.
//
// Ownership.  A request OWNS its query (copied at submission: 4 * dim bytes) and holds a reference on its filter when that
// is a device-resident FilterSet (filter_set.hpp); results are written to the caller's buffers only by whoever wins the
// request's state word (kInBatch -> kCompleting).  A caller therefore need not outlive its batch: it may LEAVE.  (A raw
// host bitmap is still the caller's memory, read by the runner while it uploads the batch: such a request is "pinned" and
// keeps the r04 behaviour -- it can leave only while it is still queued.)
//
// Cancellation.  A request whose token is up when its batch is formed is answered at once without a search
// (VK_ERR_CANCELLED for HNSW without partial results, vector_hnsw.cc:327-329; else an empty answer).  A BLOCKING caller
// polls its own token every 100 us while it waits and leaves as soon as it is up -- queued or on the device.  For
// submitted requests the watcher thread does the same every 200 us for every member of every batch in flight: the member's
// callback fires with the cancelled answer while the batch runs on, and the member's word in SearchRequest::member_cancel
// goes up, which stops the HNSW wave working on that query (the reference polls the token inside the search:
// hnswalg.h:400-402, bruteforce.h:129).  When every member of a batch is cancelled the batch's own word
// (SearchRequest::cancel_flag) goes up and the kernels stop within a millisecond.  A timed-out FT.SEARCH returns within
// ~0.2 ms of its token instead of after its batch (r04: up to a whole batch late).
//
// FLAT batches and partial fills.  One pass over the rows costs the same for 1 query or for 256, so a FLAT index gains
// nothing from a second, half-empty batch behind the one in flight: while a batch is on the device a lane is taken only
// when a FULL batch is queued (r04 with 256 blocking callers: two batches of 125 in flight, 0.47 of the device rate).  An
// HNSW batch costs per query; its lanes are taken as they come.  Filtered FLAT requests are grouped into lanes by filter
// (one scan serves every query of a lane; vk_index.cc used to send each to its own scan, or -- through submit -- N of them
// through N serial scans of one runner).
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
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "index.hpp"

namespace vk {

typedef void (*SearchDoneFn)(void *user, int status);   // == vk_search_done_fn (include/vk_index.h)

// The completer threads are the PROCESS's, not an index's: a deployment holds dozens of indexes (one per vector field) and
// the reference has one reader pool for all of them.  A piece of work is a closure; the pool grows to the largest number of
// threads any index asked for (option completer-threads) and never shrinks; its threads are never joined (an index waits
// for its own pieces, see Dispatcher::shutdown).
class CompleterPool {
 public:
  static CompleterPool &instance() {
    static CompleterPool *p = new CompleterPool();   // (never destroyed: indexes may be destroyed during static destruction)
    return *p;
  }
  template <class It>
  void post(It first, It last, uint32_t want_threads) {
    std::lock_guard<std::mutex> lk(mu_);
    while (threads_ < want_threads && threads_ < 16) {
      std::thread([this] { loop(); }).detach();
      threads_ += 1;
    }
    for (; first != last; ++first) q_.push_back(std::move(*first));
    cv_.notify_all();
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return !q_.empty(); });
      std::function<void()> f = std::move(q_.front());
      q_.pop_front();
      lk.unlock();
      f();
      f = nullptr;
      lk.lock();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  uint32_t threads_ = 0;
};

class Dispatcher {
 public:
  explicit Dispatcher(Index *ix) : ix_(ix), flat_(ix->params().algo == VK_ALGO_FLAT), dim_(ix->params().dim) {}
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
  // completer threads (0 = answers are handed out by the runner that ran the batch) and members per piece of hand-out work
  void set_completers(uint32_t threads, uint32_t chunk) {
    completer_threads_.store(std::min<uint32_t>(threads, 16), std::memory_order_relaxed);
    handout_chunk_.store(std::max<uint32_t>(chunk, 16), std::memory_order_relaxed);
  }
  bool enabled() const { return max_batch_.load(std::memory_order_relaxed) > 1; }
  uint64_t batches() const { return batches_.load(std::memory_order_relaxed); }
  uint64_t queries() const { return queries_.load(std::memory_order_relaxed); }
  uint64_t submitted() const { return submitted_.load(std::memory_order_relaxed); }
  uint64_t rejected() const { return rejected_.load(std::memory_order_relaxed); }
  uint64_t queued() const { return queued_.load(std::memory_order_relaxed); }
  uint64_t max_in_flight_seen() const { return max_active_seen_.load(std::memory_order_relaxed); }
  uint64_t left_early() const { return left_early_.load(std::memory_order_relaxed); }
  // where the runner threads' time went, summed over the runners (microseconds): waiting for a lane, inside the batching
  // window, inside Index::search (upload + kernels + download; two runners overlap on the device), handing answers out
  // on the runner itself -- and the completer threads' time inside the callers' callbacks
  struct Times { uint64_t idle_us, window_us, search_us, handout_us, completer_us; };
  Times times() const {
    return Times{t_idle_.load(std::memory_order_relaxed) / 1000, t_window_.load(std::memory_order_relaxed) / 1000,
                 t_search_.load(std::memory_order_relaxed) / 1000, t_handout_.load(std::memory_order_relaxed) / 1000,
                 t_completer_.load(std::memory_order_relaxed) / 1000};
  }

  // vk_index_set_batch_completion: with a hook set, the completions of submitted requests are told through IT -- one call per
  // piece of a finished batch (`users` / `statuses` of its members; a request answered alone, e.g. when its token goes up, is a
  // call with one member) -- and the per-request callbacks are not called.  Set before requests are submitted.
  using BatchDoneFn = void (*)(void *hook_user, void *const *users, const int *statuses, uint64_t n);
  void set_batch_done(BatchDoneFn fn, void *hook_user) {
    batch_user_ = hook_user;
    batch_fn_.store(fn, std::memory_order_release);
  }

  // vk_index_search_submit: queue one single-query request and return.  `done(user, status)` is called exactly once, from
  // a dispatcher thread, after the outputs have been written (status = the vk_status of the batch the request travelled
  // in, or of the request alone).  The query is copied; the output buffers, the token and a raw host bitmap must stay
  // valid until then (a FilterSet is kept alive by the request).
  Status submit(const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                const std::shared_ptr<FilterSet> &filter, const volatile int *cancel_flag, bool partial_ok, float *out_dist,
                uint64_t *out_label, uint64_t *out_n, SearchDoneFn done, void *user) {
    auto r = std::make_shared<Req>();
    fill(*r, query, k, ef, allow_bits, allow_nbits, filter, cancel_flag, partial_ok, out_dist, out_label, out_n);
    r->cb = done;
    r->user = user;
    return enqueue(r, /*bounded=*/true);
  }

  // vk_index_search with coalescing on: the same queue, the caller waits for its answer -- or for its token.
  Status search(const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
                const std::shared_ptr<FilterSet> &filter, const volatile int *cancel_flag, bool partial_ok, float *out_dist,
                uint64_t *out_label, uint64_t *out_n) {
    if (cancel_raised(cancel_flag)) return cancelled_answer(partial_ok, out_n);
    auto r = std::make_shared<Req>();
    fill(*r, query, k, ef, allow_bits, allow_nbits, filter, cancel_flag, partial_ok, out_dist, out_label, out_n);
    VK_TRY(enqueue(r, /*bounded=*/false));
    for (;;) {
      // (the shared wake word is read BEFORE the state: a completion in between changes it and the wait returns at once)
      const uint32_t seq = wake_seq_.load(std::memory_order_acquire);
      const uint32_t s = r->state.load(std::memory_order_acquire);
      if (s == kDone) break;
      if (cancel_raised(cancel_flag)) {
        if (s == kQueued) {
          // still QUEUED: taken out of its lane (state changes under mu_)
          std::unique_lock<std::mutex> lk(mu_);
          if (r->state.load(std::memory_order_relaxed) == kQueued) {
            auto it = lanes_.find(r->key);
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
        if (s == kInBatch && !r->pinned) {
          // on the device: the batch owns everything it reads, and only the winner of the state word writes the caller's
          // buffers -- the caller leaves, its slot of the batch's answer is dropped
          // (under the watcher's lock: the watcher reads the tokens of the members that are still kInBatch, and this caller's
          //  token dies with the call)
          bool left;
          {
            std::lock_guard<std::mutex> wl(wmu_);
            uint32_t exp = kInBatch;
            left = r->state.compare_exchange_strong(exp, kAbandoned, std::memory_order_acq_rel);
          }
          if (left) {
            left_early_.fetch_add(1, std::memory_order_relaxed);
            return cancelled_answer(partial_ok, out_n);
          }
          continue;   // (being completed right now: kDone follows within microseconds)
        }
      }
      sleepers_.fetch_add(1, std::memory_order_acq_rel);
      futex_wait(&wake_seq_, seq, cancel_flag && s != kCompleting ? 100 : 2000);
      sleepers_.fetch_sub(1, std::memory_order_relaxed);
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
    {   // (the completer threads are the process's: what they still hold of THIS index is waited for)
      std::unique_lock<std::mutex> lk(cmu_);
      ccv_.wait(lk, [&] { return pieces_out_ == 0; });
    }
    {
      std::lock_guard<std::mutex> lk(wmu_);
      stop_watch_.store(true, std::memory_order_relaxed);
    }
    wcv_.notify_all();
    if (watcher_.joinable()) watcher_.join();
  }

 private:
  enum : uint32_t { kQueued = 0, kInBatch = 1, kDone = 2, kAbandoned = 3, kCompleting = 4 };
  // a lane: requests that can share one device batch -- equal (k, ef); on a FLAT index also equal filter
  struct Key {
    uint64_t k, ef, fid;
    bool operator<(const Key &o) const { return k != o.k ? k < o.k : (ef != o.ef ? ef < o.ef : fid < o.fid); }
  };
  struct Req {
    std::vector<float> q;                 // the query, owned by the request
    const uint64_t *allow = nullptr;      // raw host bitmap (the caller's memory: `pinned`)
    uint64_t allow_nbits = 0;
    std::shared_ptr<FilterSet> filter;    // ... or a device-resident filter, kept alive by the request
    bool pinned = false;
    Key key{0, 0, 0};
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
    size_t sweep_pos = 0;          // where the watcher's look at the queued tokens goes on (sweep_queued)
    std::chrono::steady_clock::time_point last_arrival{};
  };
  // a batch on the device with at least one token among its members
  struct Watched {
    std::vector<std::shared_ptr<Req>> members;   // in batch order
    std::vector<uint32_t> words;                 // SearchRequest::member_cancel
    int word = 0;                                // SearchRequest::cancel_flag: every member is cancelled
    bool hnsw = false;
    size_t cursor = 0, n_up = 0;                 // the watcher's round robin over the members; members seen cancelled / gone
  };

  void fill(Req &r, const float *query, uint64_t k, uint64_t ef, const uint64_t *allow_bits, uint64_t allow_nbits,
            const std::shared_ptr<FilterSet> &filter, const volatile int *cancel_flag, bool partial_ok, float *od, uint64_t *ol,
            uint64_t *on) const {
    r.q.assign(query, query + dim_);
    r.filter = filter;
    r.allow = filter ? nullptr : allow_bits;
    r.allow_nbits = filter ? 0 : allow_nbits;
    r.pinned = r.allow != nullptr;
    // (bit 63 tells a bitmap's address from a FilterSet's id; HNSW serves a filter per query inside one launch)
    const uint64_t fid = !flat_ ? 0 : (filter ? filter->id() : (r.allow ? (reinterpret_cast<uint64_t>(r.allow) >> 3) | (1ull << 63) : 0));
    r.key = Key{k, ef, fid};
    r.cancel = cancel_flag;
    r.partial_ok = partial_ok;
    r.od = od;
    r.ol = ol;
    r.on = on;
  }
  Status cancelled_status(bool partial_ok) const {
    const bool hnsw = ix_->params().algo == VK_ALGO_HNSW;
    return hnsw && !partial_ok ? Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout") : Status::Ok();
  }
  Status cancelled_answer(bool partial_ok, uint64_t *out_n) const {
    *out_n = 0;
    return cancelled_status(partial_ok);
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
    Lane &lane = lanes_[r->key];
    lane.q.push_back(r);
    lane.last_arrival = r->t_submit;
    const uint64_t nq = queued_.fetch_add(1, std::memory_order_relaxed) + 1;
    submitted_.fetch_add(1, std::memory_order_relaxed);
    while (runners_.size() < in_flight_) runners_.emplace_back([this] { run(); });
    // Wake the runners only when one of them has something to decide: the batch being collected became full, or an idle
    // runner could take this lane (FLAT behind a batch in flight: only a full one).  A notify per arrival made 256 returning
    // callers wake both runners 256 times, each time through the mutex the callers are queueing on.
    const bool full = lane.q.size() >= batch_cap();
    (void)nq;
    if (lane.collector ? full : (idle_runners_ > 0 && (full || !(flat_ && active_ > 0)))) cv_.notify_all();
    return Status::Ok();
  }

  // a lane that has requests and nobody collecting from it: the one whose head has waited longest.  FLAT: while a batch is
  // on the device only a lane that can fill a whole batch (see the header)
  std::map<Key, Lane>::iterator pick_lane() {
    auto best = lanes_.end();
    const bool need_full = flat_ && active_ > 0;
    // (... unless its head has waited for the length of a pass already: a lane that never fills -- another k, a filter of its
    //  own -- must not starve behind a lane that keeps the device busy)
    const auto stale = std::chrono::steady_clock::now() - std::chrono::microseconds(std::max<uint32_t>(2000, 4 * max_wait_us_));
    for (auto it = lanes_.begin(); it != lanes_.end(); ++it) {
      if (it->second.q.empty() || it->second.collector) continue;
      if (need_full && it->second.q.size() < batch_cap() && it->second.q.front()->t_submit > stale) continue;
      if (best == lanes_.end() || it->second.q.front()->t_submit < best->second.q.front()->t_submit) best = it;
    }
    return best;
  }

  void run() {
    std::vector<std::shared_ptr<Req>> batch;
    Scratch sc;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      auto it = lanes_.end();
      idle_runners_ += 1;
      const auto t_a = std::chrono::steady_clock::now();
      cv_.wait(lk, [&] {
        if (active_ < in_flight_) it = pick_lane();
        return it != lanes_.end() || (stop_ && queued_.load(std::memory_order_relaxed) == 0);
      });
      idle_runners_ -= 1;
      if (it == lanes_.end()) return;   // stopping and drained
      const auto t_b = std::chrono::steady_clock::now();
      t_idle_.fetch_add(ns_between(t_a, t_b), std::memory_order_relaxed);
      const Key key = it->first;
      Lane *lane = &it->second;
      lane->collector = true;
      // the batching window (see the header)
      // FLAT with company around (the previous batch had several members): the callers of the batch that has just FINISHED
      // are being woken and come back within the window -- it is counted from that completion, not only from the head's
      // submission (whose own window ran out while the previous pass was on the device: with 256 blocking callers the
      // stragglers of one round and the early birds of the next otherwise settle into two half batches that take turns).
      // The pass costs the same with or without them; the wait is bounded by max_wait_us and ends early when a quarter of
      // it passes without an arrival.
      const bool company = flat_ && recent_batch_ > 1;
      const uint32_t quiet_us = company ? std::max<uint32_t>(50, max_wait_us_ / 4) : std::min<uint32_t>(200, std::max<uint32_t>(20, max_wait_us_ / 4));
      const auto quiet = std::chrono::microseconds(quiet_us);
      while (!stop_ && !lane->q.empty() && lane->q.size() < batch_cap()) {
        const auto now = std::chrono::steady_clock::now();
        auto from = lane->q.front()->t_submit;
        if (company && last_completion_ > from) from = last_completion_;
        const auto deadline = from + std::chrono::microseconds(max_wait_us_);
        if (now >= deadline) break;
        const auto since = now - std::max(lane->last_arrival, company ? last_completion_ : lane->last_arrival);
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
      if (!batch.empty()) recent_batch_ = (uint32_t)batch.size();
      lane->collector = false;
      if (lane->q.empty()) lanes_.erase(key);   // the map does not grow by one entry per (k, ef) pair ever seen
      else cv_.notify_all();                    // more than a batch was queued: another runner may start on the rest
      t_window_.fetch_add(ns_between(t_b, std::chrono::steady_clock::now()), std::memory_order_relaxed);
      if (batch.empty()) continue;              // (every queued request of the lane left, cancelled)
      active_ += 1;
      if (active_ > max_active_seen_.load(std::memory_order_relaxed)) max_active_seen_.store(active_, std::memory_order_relaxed);
      lk.unlock();
      run_batch(key.k, key.ef, batch, sc);
      batch.clear();
      lk.lock();
      active_ -= 1;
      last_completion_ = std::chrono::steady_clock::now();
      cv_.notify_all();
    }
  }

  // The one place a request is answered: whoever moves its state word kInBatch -> kCompleting OWNS it -- the caller cannot
  // leave any more (it waits for kDone), so its token and its buffers are safe to touch -- and then delivers: writes the
  // caller's buffers, fires the callback or marks it done.  claim() == false: somebody else did (the caller left, or the
  // watcher answered it when its token went up).
  static bool claim(Req &r) {
    uint32_t exp = kInBatch;
    return r.state.compare_exchange_strong(exp, kCompleting, std::memory_order_acq_rel);
  }
  // completions of submitted requests that are told in ONE call (set_batch_done): the members of a piece of a finished batch
  struct Bulk {
    BatchDoneFn hook = nullptr;   // read ONCE per piece: what was collected for it is told through it, whatever happens to the setting meanwhile
    std::vector<void *> users;
    std::vector<int> codes;
  };
  void deliver(Req &r, const Status &st, const float *d, const uint64_t *l, uint64_t n, Bulk *bulk = nullptr) {
    *r.on = st.ok() ? n : 0;
    if (st.ok() && n) {
      memcpy(r.od, d, (size_t)n * 4);
      memcpy(r.ol, l, (size_t)n * 8);
    }
    if (r.cb) {
      r.state.store(kDone, std::memory_order_release);
      BatchDoneFn hook = bulk ? bulk->hook : batch_fn_.load(std::memory_order_acquire);
      if (hook && bulk) {
        bulk->users.push_back(r.user);
        bulk->codes.push_back(st.code);
      } else if (hook) {
        void *u = r.user;
        const int c = st.code;
        hook(batch_user_, &u, &c, 1);
      } else {
        r.cb(r.user, st.code);
      }
    } else {
      r.st = st;
      r.state.store(kDone, std::memory_order_release);   // (the sleeper is woken by wake_blocked(), once per batch)
    }
  }
  bool finish(Req &r, const Status &st, const float *d, const uint64_t *l, uint64_t n) {
    if (!claim(r)) return false;
    deliver(r, st, d, l, n);
    return true;
  }
  // Blocking callers all sleep on ONE word: a batch's callers are woken by one system call instead of one each (256 wake
  // calls in a row took the runner about a millisecond, and the callers came back spread over that millisecond -- into the
  // batching window of the NEXT batch).  Callers of another batch in flight wake too, see their request is not done and go
  // back to sleep.
  void wake_blocked() {
    wake_seq_.fetch_add(1, std::memory_order_acq_rel);
    if (sleepers_.load(std::memory_order_acquire) != 0) futex_wake(&wake_seq_);
  }

  struct Scratch {
    std::vector<float> D;
    std::vector<uint64_t> L, N, nbits;
    std::vector<const float *> qtab;
    std::vector<const uint64_t *> atab;
    std::vector<const FilterSet *> ftab;
  };

  static constexpr uint64_t kOffloadMin = 32;   // (a couple of callbacks are not worth a thread hand-over)
  struct Done {   // a finished batch whose answers the completer threads hand out, piece by piece
    std::vector<std::shared_ptr<Req>> batch;
    Scratch sc;
    Status st;
    std::vector<Status> each;
    uint64_t k = 0;
    bool hnsw = false;
  };
  static uint64_t ns_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return b > a ? (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() : 0;
  }

  Status search_members(uint64_t k, uint64_t ef, const std::vector<std::shared_ptr<Req>> &batch, size_t first, size_t nq, Scratch &sc,
                        const volatile int *batch_word, const volatile uint32_t *member_words) {
    sc.qtab.resize(nq);
    bool any_raw = false, any_set = false;
    for (size_t i = 0; i < nq; ++i) {
      const Req &r = *batch[first + i];
      sc.qtab[i] = r.q.data();
      any_raw = any_raw || r.allow != nullptr;
      any_set = any_set || r.filter != nullptr;
    }
    SearchRequest rq;
    rq.query_tab = sc.qtab.data();
    rq.nq = nq;
    rq.k = k;
    rq.ef = ef;
    rq.partial_ok = true;   // (per-member rule in run_batch)
    if (flat_ && (any_raw || any_set)) {   // a FLAT lane shares one filter: one scan
      const Req &r0 = *batch[first];
      rq.filter = r0.filter.get();
      rq.allow_bits = r0.allow;
      rq.allow_nbits = r0.allow_nbits;
    } else {
      if (any_raw) {
        sc.atab.resize(nq);
        sc.nbits.resize(nq);
        for (size_t i = 0; i < nq; ++i) { sc.atab[i] = batch[first + i]->allow; sc.nbits[i] = batch[first + i]->allow_nbits; }
        rq.allow_tab = sc.atab.data();
        rq.allow_nbits_tab = sc.nbits.data();
      }
      if (any_set) {
        sc.ftab.resize(nq);
        for (size_t i = 0; i < nq; ++i) sc.ftab[i] = batch[first + i]->filter.get();
        rq.filter_tab = sc.ftab.data();
      }
    }
    rq.cancel_flag = batch_word;
    rq.member_cancel = member_words;
    return ix_->search(rq, sc.D.data() + first * k, sc.L.data() + first * k, sc.N.data() + first);
  }

  void run_batch(uint64_t k, uint64_t ef, std::vector<std::shared_ptr<Req>> &batch, Scratch &sc) {
    const bool hnsw = ix_->params().algo == VK_ALGO_HNSW;
    // requests whose token is already up are answered without a search
    // (A token is read only while its request is still kInBatch and under wmu_: a blocking caller whose token went up
    //  leaves by changing the state under that lock, and its token -- a word on its stack -- dies with the call.  Members
    //  that left already are dropped here.  The answers go out after the lock is released; a member that leaves in
    //  between loses nothing, finish() fails to claim it.)
    size_t live = 0;
    bool any_token = false;
    for (size_t i = 0; i < batch.size() && !any_token; ++i) any_token = batch[i]->cancel != nullptr;
    if (any_token) {
      std::vector<std::shared_ptr<Req>> up;
      {
        std::lock_guard<std::mutex> wl(wmu_);
        for (size_t i = 0; i < batch.size(); ++i) {
          Req &r = *batch[i];
          if (r.state.load(std::memory_order_acquire) != kInBatch) continue;   // left while it was queued for a runner
          if (cancel_raised(r.cancel)) {
            up.push_back(std::move(batch[i]));
          } else {
            if (live != i) batch[live] = std::move(batch[i]);
            ++live;
          }
        }
      }
      for (auto &r : up) finish(*r, cancelled_status(r->partial_ok), nullptr, nullptr, 0);
      if (!up.empty()) wake_blocked();
      batch.resize(live);
    }
    const uint64_t nq = batch.size();
    if (nq == 0) return;
    Status st = Status::Ok();
    std::shared_ptr<Watched> w;
    std::vector<Status> each;   // per-member status when the batch had to be re-run member by member
    const auto t_s = std::chrono::steady_clock::now();
    try {
      sc.D.resize(nq * k);
      sc.L.resize(nq * k);
      sc.N.assign(nq, 0);
      if (any_token) {
        w = std::make_shared<Watched>();
        w->members = batch;
        w->words.assign(nq, 0);
        w->hnsw = hnsw;
        watch(w);
      }
      st = search_members(k, ef, batch, 0, nq, sc, w ? &w->word : nullptr, w ? w->words.data() : nullptr);
      if (!st.ok() && st.code == VK_ERR_INVALID && nq > 1) {
        // an argument error may be ONE member's (a filter built for another index, ...): the members are tried alone so
        // that it fails alone
        each.resize(nq);
        for (uint64_t i = 0; i < nq; ++i)
          each[i] = search_members(k, ef, batch, i, 1, sc, w ? &w->word : nullptr, w ? w->words.data() + i : nullptr);
      }
    } catch (const std::exception &e) {
      st = Status::Err(VK_ERR_INTERNAL, e.what());
      each.clear();
    }
    if (w) unwatch(w);
    batches_.fetch_add(1, std::memory_order_relaxed);
    queries_.fetch_add(nq, std::memory_order_relaxed);
    const auto t_h = std::chrono::steady_clock::now();
    t_search_.fetch_add(ns_between(t_s, t_h), std::memory_order_relaxed);
    // Handing the answers out -- per member a copy and a callback into the caller's code (which builds its reply and posts it
    // to a pool of its own: microseconds each, more when that pool's lock is contended) -- is not the runner's work: 8192
    // callbacks take as long as the HNSW launch that produced them, and 256 of them delayed the next FLAT pass by a quarter
    // of its length.  A batch with callbacks is cut into pieces that the completer threads hand out side by side while the
    // runner forms the next batch; a batch of blocking callers only (a copy each and ONE wake-up) is answered right here.
    const uint32_t n_completers = completer_threads_.load(std::memory_order_relaxed);
    bool any_cb = false;
    for (uint64_t i = 0; i < nq && !any_cb; ++i) any_cb = batch[i]->cb != nullptr;
    if (any_cb && n_completers != 0 && nq >= kOffloadMin) {
      auto d = std::make_shared<Done>();
      d->batch.swap(batch);
      std::swap(d->sc, sc);
      d->st = st;
      d->each.swap(each);
      d->k = k;
      d->hnsw = hnsw;
      const uint64_t chunk = std::max<uint64_t>(handout_chunk_.load(std::memory_order_relaxed), (nq + 4 * n_completers - 1) / (4 * n_completers));
      std::vector<std::function<void()>> work;
      for (uint64_t first = 0; first < nq; first += chunk) {
        const uint64_t count = std::min<uint64_t>(chunk, nq - first);
        work.emplace_back([this, d, first, count] {
          const auto t0 = std::chrono::steady_clock::now();
          hand_out(d->batch, d->sc, d->st, d->each, d->k, d->hnsw, first, count);
          t_completer_.fetch_add(ns_between(t0, std::chrono::steady_clock::now()), std::memory_order_relaxed);
          piece_done();
        });
      }
      d.reset();   // (the pieces hold the batch; the last one to finish frees it)
      {
        std::lock_guard<std::mutex> lk(cmu_);
        pieces_out_ += work.size();
      }
      CompleterPool::instance().post(work.begin(), work.end(), n_completers);
      return;
    }
    hand_out(batch, sc, st, each, k, hnsw, 0, nq);
    t_handout_.fetch_add(ns_between(t_h, std::chrono::steady_clock::now()), std::memory_order_relaxed);
  }

  void hand_out(std::vector<std::shared_ptr<Req>> &batch, Scratch &sc, const Status &st, const std::vector<Status> &each, uint64_t k, bool hnsw,
                uint64_t first, uint64_t count) {
    Bulk bulk;
    bulk.hook = batch_fn_.load(std::memory_order_acquire);
    if (bulk.hook) { bulk.users.reserve(count); bulk.codes.reserve(count); }
    for (uint64_t i = first; i < first + count; ++i) {
      Req &r = *batch[i];
      if (!claim(r)) continue;   // (it left, or was answered when its token went up; its token may be gone: not read)
      Status mine = each.empty() ? st : each[i];
      uint64_t n = mine.ok() ? sc.N[i] : 0;
      if (mine.ok() && hnsw && !r.partial_ok && cancel_raised(r.cancel)) {
        // cancelled while the batch it travelled in was on the device: the reference's answer for a raised token
        // (vector_hnsw.cc:327-329), whatever the batch found
        n = 0;
        mine = Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
      }
      deliver(r, mine, sc.D.data() + i * k, sc.L.data() + i * k, n, &bulk);
    }
    if (!bulk.users.empty())   // one call for the piece's submitted members (set_batch_done)
      bulk.hook(batch_user_, bulk.users.data(), bulk.codes.data(), bulk.users.size());
    wake_blocked();
  }
  void piece_done() {
    // (the closure -- and with it the piece's reference on the batch -- is destroyed by the pool thread AFTER this returns: the
    //  batch's requests and buffers are the batch's own, nothing of the dispatcher is touched by that)
    std::lock_guard<std::mutex> lk(cmu_);
    if (--pieces_out_ == 0) ccv_.notify_all();
  }

  // ---- the watcher: batches on the device that carry tokens ---------------------------------------------------------
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
  // Every tick (200 us) looks at up to kMemberBudget tokens of the batches in flight, round robin -- a batch of 8192 members
  // is covered in about a millisecond; the first version looked at every member of every batch every tick (16 384 cache-missing
  // loads per 200 us with two HNSW batches in flight, under the lock the runners need for watch / unwatch) -- and, every
  // fifth tick, at up to kQueuedBudget queued requests of one lane (under the queue's lock: 12 us per millisecond).
  static constexpr size_t kMemberBudget = 2048, kQueuedBudget = 512;
  void watch_loop() {
    std::unique_lock<std::mutex> lk(wmu_);
    uint32_t tick = 0;
    for (;;) {
      if (watched_.empty()) wcv_.wait(lk, [&] { return !watched_.empty() || stop_flag(); });
      else wcv_.wait_for(lk, std::chrono::microseconds(200));
      if (stop_flag() && watched_.empty()) return;
      size_t budget = kMemberBudget;
      std::vector<std::shared_ptr<Req>> mine;   // claimed under the lock, answered after it is dropped
      for (auto &w : watched_) {
        if (budget == 0) break;
        if (__atomic_load_n(&w->word, __ATOMIC_RELAXED)) continue;
        const size_t n = w->members.size();
        for (size_t step = 0; step < n && budget != 0; ++step, --budget) {
          const size_t i = w->cursor;
          w->cursor = i + 1 == n ? 0 : i + 1;
          if (__atomic_load_n(&w->words[i], __ATOMIC_RELAXED)) continue;   // (seen before: counted in n_up)
          Req &r = *w->members[i];
          // A member that is no longer kInBatch has left or was answered: its token may be gone and is NOT read (a blocking
          // caller changes the state under wmu_, which this thread holds; the runner's own completions come after unwatch)
          const bool gone = r.state.load(std::memory_order_acquire) != kInBatch;
          if (!gone && !cancel_raised(r.cancel)) continue;
          __atomic_store_n(&w->words[i], 1u, __ATOMIC_RELAXED);             // the wave working on this member stops
          w->n_up += 1;
          // ... and the member is answered now: the rest of its batch runs on without it
          if (!gone && !r.pinned && r.cb && claim(r)) mine.push_back(w->members[i]);
        }
        if (w->n_up == n) __atomic_store_n(&w->word, 1, __ATOMIC_RELAXED);   // every member is cancelled: the kernels stop
      }
      // A completion callback is the caller's code (the adaptor builds a reply and posts it on; it may search again): it runs
      // without wmu_, which every runner needs for watch / unwatch and every blocking caller to leave on its token.
      if (!mine.empty()) {
        lk.unlock();
        for (auto &r : mine) deliver(*r, cancelled_status(r->partial_ok), nullptr, nullptr, 0);
        left_early_.fetch_add(mine.size(), std::memory_order_relaxed);
        mine.clear();
        lk.lock();
      }
      // ... and the submitted requests that are still QUEUED behind the batches in flight: a token that goes up there is
      // answered now, not when a runner gets to its lane (blocking callers poll their own).  Outside wmu_: the queue has its
      // own lock, and the callbacks run without either.
      if (++tick % 5 == 0) {
        lk.unlock();
        sweep_queued();
        lk.lock();
      }
    }
  }
  void sweep_queued() {
    std::vector<std::shared_ptr<Req>> gone;
    {
      std::lock_guard<std::mutex> ql(mu_);
      if (lanes_.empty()) return;
      // one lane per call, round robin by key; within it a window that moves from the front to the back
      auto it = lanes_.upper_bound(sweep_key_);
      if (it == lanes_.end()) it = lanes_.begin();
      sweep_key_ = it->first;
      auto &q = it->second.q;
      size_t pos = it->second.sweep_pos < q.size() ? it->second.sweep_pos : 0;
      size_t budget = kQueuedBudget;
      while (pos < q.size() && budget-- != 0) {
        Req &r = *q[pos];
        if (r.cb && cancel_raised(r.cancel)) {
          r.state.store(kInBatch, std::memory_order_relaxed);   // (claimed below like a member of a batch)
          gone.push_back(std::move(q[pos]));
          q.erase(q.begin() + (std::ptrdiff_t)pos);
          queued_.fetch_sub(1, std::memory_order_relaxed);
        } else {
          ++pos;
        }
      }
      it->second.sweep_pos = pos;
      if (q.empty() && !it->second.collector) lanes_.erase(it);   // (a collector holds a pointer to its lane)
    }
    for (auto &r : gone)
      if (finish(*r, cancelled_status(r->partial_ok), nullptr, nullptr, 0)) left_early_.fetch_add(1, std::memory_order_relaxed);
  }
  bool stop_flag() const { return stop_watch_.load(std::memory_order_relaxed); }

  Index *ix_;
  const bool flat_;
  const uint32_t dim_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::map<Key, Lane> lanes_;
  Key sweep_key_{0, 0, 0};
  std::vector<std::thread> runners_;
  uint32_t in_flight_ = 2, active_ = 0, idle_runners_ = 0, recent_batch_ = 0;
  bool stop_ = false;
  std::chrono::steady_clock::time_point last_completion_{};
  std::atomic<uint32_t> max_batch_{0};
  uint32_t max_wait_us_ = 0;
  std::atomic<uint32_t> wake_seq_{0}, sleepers_{0};
  std::atomic<uint64_t> queue_depth_{100000};
  std::atomic<uint32_t> completer_threads_{6}, handout_chunk_{64};
  std::atomic<uint64_t> t_idle_{0}, t_window_{0}, t_search_{0}, t_handout_{0}, t_completer_{0};   // nanoseconds
  std::atomic<uint64_t> queued_{0}, batches_{0}, queries_{0}, submitted_{0}, rejected_{0}, max_active_seen_{0}, left_early_{0};
  std::mutex cmu_;
  std::condition_variable ccv_;
  std::atomic<BatchDoneFn> batch_fn_{nullptr};
  void *batch_user_ = nullptr;
  uint64_t pieces_out_ = 0;   // pieces of finished batches the process's completer threads still hold (under cmu_)
  std::mutex wmu_;
  std::condition_variable wcv_;
  std::vector<std::shared_ptr<Watched>> watched_;
  std::thread watcher_;
  std::atomic<bool> stop_watch_{false};
};

}  // namespace vk
