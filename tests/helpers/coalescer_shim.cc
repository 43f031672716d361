// CPU-side harness for csrc/dispatcher.hpp (N1): a fake index whose batched search answers every query with
// a function of the query alone and records the batch sizes it was called with.  Many threads issue
// single-query requests through the Dispatcher -- blocking (search) and non-blocking (submit + callback); each must
// get exactly its own answer, every request must be served exactly once, concurrent requests must actually travel
// together, and with two runners two batches must be in flight at once.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "dispatcher.hpp"

namespace {
struct FakeIndex final : vk::Index {
  explicit FakeIndex(const vk_index_params &p) : Index(p) {}
  std::atomic<uint64_t> calls{0}, queries{0}, max_batch{0}, concurrent{0}, max_concurrent{0}, cancelled_batches{0}, member_words_seen{0};
  int delay_us = 200;   // a device pass takes a while: requests pile up behind it
  vk::Status search(const vk::SearchRequest &rq, float *od, uint64_t *ol, uint64_t *on) override {
    calls += 1;
    queries += rq.nq;
    uint64_t m = max_batch.load();
    while (rq.nq > m && !max_batch.compare_exchange_weak(m, rq.nq)) {}
    const uint64_t c = concurrent.fetch_add(1) + 1;
    uint64_t mc = max_concurrent.load();
    while (c > mc && !max_concurrent.compare_exchange_weak(mc, c)) {}
    // the "device pass": like the real kernels it polls the batch's cancellation word and stops early when it goes up
    const auto end = std::chrono::steady_clock::now() + std::chrono::microseconds(delay_us);
    bool stopped = false;
    std::vector<uint8_t> seen(rq.nq, 0);
    while (std::chrono::steady_clock::now() < end) {
      if (vk::cancel_raised(rq.cancel_flag)) { stopped = true; break; }
      // (like the HNSW kernel: a member's own word stops the work on that member only)
      if (rq.member_cancel)
        for (uint64_t q = 0; q < rq.nq; ++q)
          if (!seen[q] && __atomic_load_n(const_cast<const uint32_t *>(&rq.member_cancel[q]), __ATOMIC_RELAXED)) { seen[q] = 1; member_words_seen += 1; }
      std::this_thread::sleep_for(std::chrono::microseconds(delay_us > 1000 ? 100 : 20));
    }
    if (stopped) cancelled_batches += 1;
    concurrent.fetch_sub(1);
    if (rq.k == 13) return vk::Status::Err(VK_ERR_INTERNAL, "k = 13 fails on purpose");
    for (uint64_t q = 0; q < rq.nq; ++q) {
      const float *v = rq.query_tab ? rq.query_tab[q] : rq.queries + q * params_.dim;
      const uint64_t n = rq.k < 3 ? rq.k : 3;            // "fewer than k found" is part of the contract
      on[q] = n;
      // a per-query filter shows in the answer: word 0 of the query's own bitmap is added to every label
      // (a FLAT lane shares one filter: it arrives as the batch's allow_bits)
      const uint64_t tag = rq.allow_tab && rq.allow_tab[q] ? rq.allow_tab[q][0] + rq.allow_nbits_tab[q] : (rq.allow_bits ? rq.allow_bits[0] + rq.allow_nbits : 0);
      for (uint64_t i = 0; i < n; ++i) {
        od[q * rq.k + i] = v[0] * 1000.f + (float)i + (float)rq.ef * 0.001f;
        ol[q * rq.k + i] = (uint64_t)v[1] * 10 + i + tag;
      }
    }
    return vk::Status::Ok();
  }
  // not used by the coalescer
  vk::Status add(uint64_t, const float *) override { return vk::Status::Ok(); }
  vk::Status add_batch(const uint64_t *, const float *, uint64_t) override { return vk::Status::Ok(); }
  vk::Status remove(uint64_t) override { return vk::Status::Ok(); }
  vk::Status resize(uint64_t) override { return vk::Status::Ok(); }
  vk::Status set_ef(uint32_t) override { return vk::Status::Ok(); }
  vk::Status flush() override { return vk::Status::Ok(); }
  vk::Status search_device(const vk::SearchRequest &, float *, uint64_t *, uint32_t *, hipStream_t) override { return vk::Status::Ok(); }
  vk::Status label_distances(const float *, const uint64_t *, uint64_t, float *, uint8_t *) override { return vk::Status::Ok(); }
  vk::Status distance(uint64_t, const float *, float *) override { return vk::Status::Ok(); }
  vk::Status get_row(uint64_t, float *) override { return vk::Status::Ok(); }
  vk::Status contains(uint64_t, bool *) override { return vk::Status::Ok(); }
  vk::Status stats(vk_index_stats *) override { return vk::Status::Ok(); }
  vk::Status device_rows(uint64_t, void **, uint64_t *) override { return vk::Status::Ok(); }
  vk::Status commit_device_rows(uint64_t, const uint64_t *) override { return vk::Status::Ok(); }
  vk::Status save(vk_write_chunk_fn, void *) override { return vk::Status::Ok(); }
  void filter_devices(std::vector<int> *out) const override { out->clear(); }
};
}  // namespace

// threads x per_thread requests, lanes (k, ef) taken round robin from `lanes` pairs; returns 0 when every
// request got its own answer.  out[0..3] = device calls, queries served, largest batch, coalescer batches.
extern "C" int coalescer_run(int threads, int per_thread, uint32_t max_batch, uint32_t max_wait_us, int n_lanes,
                             int with_failing_lane, uint64_t *out) {
  const bool toggle = (with_failing_lane & 2) != 0;
  with_failing_lane &= 1;
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = VK_ALGO_HNSW;   // (a filter per query inside one batch; a FLAT index groups filtered requests into lanes by filter: dispatcher_async_run)
  FakeIndex ix(p);
  vk::Dispatcher co(&ix);
  co.configure(max_batch, max_wait_us);
  std::atomic<int> bad{0};
  auto worker = [&](int t) {
    for (int r = 0; r < per_thread; ++r) {
      const int id = t * per_thread + r;
      const int lane = id % n_lanes;
      const bool fail = with_failing_lane && lane == n_lanes - 1;
      const uint64_t k = fail ? 13 : (uint64_t)(lane + 1);      // k = 1, 2, 3, ... (answers carry min(k,3) entries)
      const uint64_t ef = 100 + (uint64_t)lane;
      float q[4] = {(float)id, (float)(id + 7), 0.f, 0.f};
      float d[16];
      uint64_t l[16], n = 99;
      // every third request carries its own filter; every 17th is cancelled before it is served and must come back
      // at once with nothing, whatever the batch it would have ridden in does
      const bool filtered = id % 3 == 0, cancelled = !fail && id % 17 == 5;
      const uint64_t bits[1] = {(uint64_t)id * 1000};
      const uint64_t nbits = 64 + (uint64_t)(id % 5);
      volatile int flag = cancelled ? 1 : 0;
      if (toggle && id == threads * per_thread / 2) co.configure(0, 0);      // coalescing switched off under load
      vk::Status st = co.search(q, k, ef, filtered ? bits : nullptr, filtered ? nbits : 0, nullptr, &flag, true, d, l, &n);
      if (cancelled) {
        if (!st.ok() || n != 0) bad += 1;
        continue;
      }
      if (fail) {
        if (st.ok() || st.msg.find("on purpose") == std::string::npos) bad += 1;
        continue;
      }
      const uint64_t want_n = k < 3 ? k : 3;
      if (!st.ok() || n != want_n) { bad += 1; continue; }
      for (uint64_t i = 0; i < n; ++i)
        if (d[i] != (float)id * 1000.f + (float)i + (float)ef * 0.001f ||
            l[i] != (uint64_t)(id + 7) * 10 + i + (filtered ? bits[0] + nbits : 0)) bad += 1;
    }
  };
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t) ts.emplace_back(worker, t);
  for (auto &t : ts) t.join();
  out[0] = ix.calls;
  out[1] = ix.queries;
  out[2] = ix.max_batch;
  out[3] = co.batches();
  out[4] = co.queries();
  return bad.load();
}

// ---- the non-blocking entry: `producers` threads submit `per_producer` requests each without waiting (up to
// `window` outstanding per producer), completions arrive through the callback.  Checks: every accepted request completes
// exactly once with its own answer; a full queue rejects with VK_ERR_BUSY and that request's callback never fires; with
// two runners two "device passes" overlap.  out[0..5] = device calls, largest batch, most concurrent passes, rejected,
// completions, cancelled completions.
namespace {
struct AsyncSlot {
  float q[4];
  float d[16];
  uint64_t l[16], n;
  uint64_t bits[1];
  volatile int flag;
  std::atomic<int> done{0};
  int status = -1;
  int id = 0;
  bool filtered = false, cancelled = false;
  std::atomic<uint64_t> *completions;
};
void async_done(void *user, int status) {
  AsyncSlot *s = static_cast<AsyncSlot *>(user);
  s->status = status;
  s->completions->fetch_add(1);
  s->done.fetch_add(1, std::memory_order_release);
}
// completions told in bulk (Dispatcher::set_batch_done): the per-request callback must then never be called
std::atomic<int> g_bulk_mode{0};
std::atomic<uint64_t> g_bulk_calls{0}, g_bulk_max_span{0}, g_stray_callbacks{0};
void never_done(void *, int) { g_stray_callbacks.fetch_add(1); }
void bulk_done(void *, void *const *users, const int *statuses, uint64_t n) {
  g_bulk_calls.fetch_add(1);
  uint64_t m = g_bulk_max_span.load();
  while (n > m && !g_bulk_max_span.compare_exchange_weak(m, n)) {}
  for (uint64_t i = 0; i < n; ++i) async_done(users[i], statuses[i]);
}
}  // namespace

// the next dispatcher_async_run tells its completions through the bulk hook; out[0..2] of dispatcher_bulk_stats: hook calls,
// the largest span, per-request callbacks that fired although the hook was set (must be 0)
extern "C" void dispatcher_async_bulk(int on) { g_bulk_mode = on; g_bulk_calls = 0; g_bulk_max_span = 0; g_stray_callbacks = 0; }
extern "C" void dispatcher_bulk_stats(uint64_t *out) { out[0] = g_bulk_calls; out[1] = g_bulk_max_span; out[2] = g_stray_callbacks; }

extern "C" int dispatcher_async_run(int producers, int per_producer, int window, uint32_t max_batch, uint32_t max_wait_us,
                                    uint32_t in_flight, uint64_t queue_depth, int delay_us, int hnsw, uint64_t *out) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = hnsw ? VK_ALGO_HNSW : VK_ALGO_FLAT;
  FakeIndex ix(p);
  ix.delay_us = delay_us;
  std::atomic<int> bad{0};
  std::atomic<uint64_t> completions{0}, rejected{0}, cancelled_done{0};
  {
    vk::Dispatcher dp(&ix);
    dp.configure(max_batch, max_wait_us);
    dp.set_in_flight(in_flight);
    dp.set_queue_depth(queue_depth);
    const bool bulk = g_bulk_mode.load() != 0;
    if (bulk) dp.set_batch_done(&bulk_done, nullptr);
    auto producer = [&](int t) {
      std::vector<std::unique_ptr<AsyncSlot>> slots;
      size_t checked = 0;
      auto check = [&](AsyncSlot &s) {
        while (s.done.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        if (s.done.load() != 1) bad += 1;
        const uint64_t k = 3, ef = 100;
        if (s.cancelled) {   // token up before the batch formed: answered without a search
          cancelled_done += 1;
          if (hnsw ? s.status != VK_ERR_CANCELLED : (s.status != VK_OK || s.n != 0)) bad += 1;
          return;
        }
        if (s.status != VK_OK || s.n != 3) { bad += 1; return; }
        for (uint64_t i = 0; i < s.n; ++i)
          if (s.d[i] != (float)s.id * 1000.f + (float)i + (float)ef * 0.001f ||
              s.l[i] != (uint64_t)(s.id + 7) * 10 + i + (s.filtered ? s.bits[0] + 64 : 0)) bad += 1;
        (void)k;
      };
      for (int r = 0; r < per_producer; ++r) {
        auto s = std::make_unique<AsyncSlot>();
        s->id = t * per_producer + r;
        s->q[0] = (float)s->id; s->q[1] = (float)(s->id + 7); s->q[2] = s->q[3] = 0.f;
        s->filtered = s->id % 3 == 0;
        s->cancelled = s->id % 29 == 11;
        s->bits[0] = (uint64_t)s->id * 1000;
        s->flag = s->cancelled ? 1 : 0;
        s->n = 99;
        s->completions = &completions;
        vk::Status st = dp.submit(s->q, 3, 100, s->filtered ? s->bits : nullptr, s->filtered ? 64 : 0, nullptr, s->id % 2 ? &s->flag : (s->cancelled ? &s->flag : nullptr),
                                  /*partial_ok=*/false, s->d, s->l, &s->n, bulk ? never_done : async_done, s.get());
        if (!st.ok()) {
          if (st.code != VK_ERR_BUSY) bad += 1;
          rejected += 1;
          std::this_thread::sleep_for(std::chrono::microseconds(50));
          // (a rejected request's callback must never fire: its slot is checked for done == 0 at the end)
          s->id = -1;
        }
        slots.push_back(std::move(s));
        while (slots.size() - checked > (size_t)window) {
          if (slots[checked]->id >= 0) check(*slots[checked]);
          ++checked;
        }
      }
      for (; checked < slots.size(); ++checked)
        if (slots[checked]->id >= 0) check(*slots[checked]);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      for (auto &s : slots)
        if (s->id < 0 && s->done.load() != 0) bad += 1;
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < producers; ++t) ts.emplace_back(producer, t);
    for (auto &t : ts) t.join();
    if (dp.rejected() != rejected.load()) bad += 1;
    if (dp.submitted() != completions.load()) bad += 1;
    out[2] = dp.max_in_flight_seen();
    // (bits 0-31: the completer threads' time inside callbacks, bits 32-63: what the runners handed out themselves; us)
    const vk::Dispatcher::Times tm = dp.times();
    out[7] = (std::min<uint64_t>(tm.handout_us, 0xFFFFFFFFull) << 32) | std::min<uint64_t>(tm.completer_us, 0xFFFFFFFFull);
  }   // (the dispatcher is destroyed with nothing queued)
  out[0] = ix.calls;
  out[1] = ix.max_batch;
  out[3] = rejected;
  out[4] = completions;
  out[5] = cancelled_done;
  out[6] = ix.max_concurrent;
  return bad.load();
}

// destroy with requests still queued and batches on the "device": every accepted request's callback fires exactly once
// before the destructor returns, nothing is touched afterwards (ASAN), and no runner outlives the dispatcher (TSAN).
extern "C" int dispatcher_destroy_run(int rounds, int n_req) {
  std::atomic<int> bad{0};
  for (int r = 0; r < rounds; ++r) {
    vk_index_params p{};
    p.struct_size = sizeof p;
    p.dim = 4;
    FakeIndex ix(p);
    ix.delay_us = 1500;
    std::atomic<uint64_t> completions{0};
    std::vector<std::unique_ptr<AsyncSlot>> slots;
    uint64_t accepted = 0;
    {
      vk::Dispatcher dp(&ix);
      dp.configure(8, 5000);
      dp.set_in_flight(2);
      for (int i = 0; i < n_req; ++i) {
        auto s = std::make_unique<AsyncSlot>();
        s->id = i;
        s->q[0] = (float)i; s->q[1] = (float)(i + 7); s->q[2] = s->q[3] = 0.f;
        s->flag = 0;
        s->completions = &completions;
        if (dp.submit(s->q, 3, 100 + (uint64_t)(i % 2), nullptr, 0, nullptr, nullptr, true, s->d, s->l, &s->n, async_done, s.get()).ok()) accepted += 1;
        slots.push_back(std::move(s));
      }
      if (r % 2) std::this_thread::sleep_for(std::chrono::microseconds(700));   // some batches are on the device by now
    }   // ~Dispatcher: answers what is queued, waits for the callbacks
    if (completions.load() != accepted) bad += 1;
    for (auto &s : slots)
      if (s->done.load() != 1 || s->status != VK_OK || s->n != 3 || s->l[0] != (uint64_t)(s->id + 7) * 10) bad += 1;
  }
  return bad.load();
}

// Two indexes share the process's completer threads (CompleterPool): both serve submissions whose batches are handed out
// in pieces by the pool; index A is destroyed while B's pieces are still queued and running -- A's destructor waits for A's
// own pieces only, B goes on, and nothing of A is touched afterwards (ASAN / TSAN).
extern "C" int dispatcher_two_indexes_run(int rounds, int n_req) {
  std::atomic<int> bad{0};
  for (int r = 0; r < rounds; ++r) {
    vk_index_params p{};
    p.struct_size = sizeof p;
    p.dim = 4;
    p.algo = VK_ALGO_HNSW;
    FakeIndex ia(p), ib(p);
    ia.delay_us = 300;
    ib.delay_us = 900;
    std::atomic<uint64_t> ca{0}, cb{0};
    std::vector<std::unique_ptr<AsyncSlot>> sa, sb;
    uint64_t acc_a = 0, acc_b = 0;
    auto feed = [&](vk::Dispatcher &dp, std::vector<std::unique_ptr<AsyncSlot>> &slots, std::atomic<uint64_t> &cnt, uint64_t &acc) {
      for (int i = 0; i < n_req; ++i) {
        auto s = std::make_unique<AsyncSlot>();
        s->id = i;
        s->q[0] = (float)i; s->q[1] = (float)(i + 7); s->q[2] = s->q[3] = 0.f;
        s->flag = 0;
        s->completions = &cnt;
        if (dp.submit(s->q, 3, 100, nullptr, 0, nullptr, nullptr, true, s->d, s->l, &s->n, async_done, s.get()).ok()) acc += 1;
        slots.push_back(std::move(s));
      }
    };
    {
      vk::Dispatcher db(&ib);
      db.configure(64, 2000);
      db.set_in_flight(2);
      db.set_completers(3, 16);
      {
        vk::Dispatcher da(&ia);
        da.configure(64, 2000);
        da.set_in_flight(2);
        da.set_completers(2, 16);
        std::thread tb([&] { feed(db, sb, cb, acc_b); });
        feed(da, sa, ca, acc_a);
        tb.join();
      }   // ~Dispatcher A: B still has batches on its "device" and pieces in the pool
      if (ca.load() != acc_a) bad += 1;
      for (auto &s : sa)
        if (s->done.load() != 1 || s->status != VK_OK || s->n != 3 || s->l[0] != (uint64_t)(s->id + 7) * 10) bad += 1;
    }
    if (cb.load() != acc_b) bad += 1;
    for (auto &s : sb)
      if (s->done.load() != 1 || s->status != VK_OK || s->n != 3 || s->l[0] != (uint64_t)(s->id + 7) * 10) bad += 1;
  }
  return bad.load();
}

// Every member of a batch carries a token and all of them go up while the batch is on the device: the batch's own
// cancellation word must go up (the fake device pass polls it like the kernels do) and the callers come back long before
// the pass would have ended; with one live member among them the batch must run to its end.
extern "C" int dispatcher_batch_cancel_run(int all_cancel, uint64_t *out) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = VK_ALGO_HNSW;
  FakeIndex ix(p);
  ix.delay_us = 300000;   // 0.3 s
  vk::Dispatcher dp(&ix);
  dp.configure(8, 20000);
  std::atomic<int> bad{0};
  std::vector<std::thread> ts;
  volatile int flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < 8; ++t)
    ts.emplace_back([&, t] {
      float q[4] = {(float)t, (float)(t + 7), 0.f, 0.f}, d[16];
      uint64_t l[16], n = 99;
      vk::Status st = dp.search(q, 3, 100, nullptr, 0, nullptr, &flags[t], /*partial_ok=*/false, d, l, &n);
      const bool mine_up = all_cancel || t != 0;
      if (mine_up ? st.code != VK_ERR_CANCELLED : (!st.ok() || n != 3)) bad += 1;
    });
  std::this_thread::sleep_for(std::chrono::milliseconds(40));     // the batch of eight is on the device
  for (int t = all_cancel ? 0 : 1; t < 8; ++t) __atomic_store_n(const_cast<int *>(&flags[t]), 1, __ATOMIC_RELAXED);
  for (auto &t : ts) t.join();
  out[0] = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  // (the callers leave on their own tokens; the batch they left behind stops when the watcher has seen all eight)
  dp.shutdown();
  out[3] = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  out[1] = ix.cancelled_batches;
  out[2] = ix.calls;
  return bad.load();
}

// ONE member of a batch on the device is cancelled while the others live on: it must come back at once (r04: with its batch).
// Blocking callers poll their own token; submitted requests are answered by the watcher.  The batch runs to its end, the
// member's own word goes up (the HNSW kernel's wave stops there), the other members get their full answers.
// out[0] = microseconds between raising the member's token and its return / callback, out[1] = total milliseconds,
// out[2] = member words the "device" saw go up, out[3] = dispatcher's count of early leavers.
namespace {
struct CancelSlot {
  float q[4], d[16];
  uint64_t l[16], n = 99;
  volatile int flag = 0;
  std::atomic<int> done{0};
  int status = -1;
  std::chrono::steady_clock::time_point t_done;
};
void cancel_done(void *user, int status) {
  CancelSlot *s = static_cast<CancelSlot *>(user);
  s->status = status;
  s->t_done = std::chrono::steady_clock::now();
  s->done.store(1, std::memory_order_release);
}
}  // namespace

extern "C" int dispatcher_member_cancel_run(int hnsw, int use_submit, int victim, uint64_t *out) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = hnsw ? VK_ALGO_HNSW : VK_ALGO_FLAT;
  FakeIndex ix(p);
  ix.delay_us = 200000;   // 0.2 s on the "device"
  std::atomic<int> bad{0};
  std::vector<std::unique_ptr<CancelSlot>> slots;
  for (int t = 0; t < 8; ++t) {
    slots.push_back(std::make_unique<CancelSlot>());
    slots[t]->q[0] = (float)t; slots[t]->q[1] = (float)(t + 7); slots[t]->q[2] = slots[t]->q[3] = 0.f;
  }
  std::chrono::steady_clock::time_point t_raise, t_back[8];
  const auto t0 = std::chrono::steady_clock::now();
  {
    vk::Dispatcher dp(&ix);
    dp.configure(8, 20000);
    std::vector<std::thread> ts;
    if (use_submit) {
      for (int t = 0; t < 8; ++t)
        if (!dp.submit(slots[t]->q, 3, 100, nullptr, 0, nullptr, &slots[t]->flag, /*partial_ok=*/false, slots[t]->d, slots[t]->l, &slots[t]->n,
                       cancel_done, slots[t].get()).ok()) bad += 1;
    } else {
      for (int t = 0; t < 8; ++t)
        ts.emplace_back([&, t] {
          CancelSlot &s = *slots[t];
          vk::Status st = dp.search(s.q, 3, 100, nullptr, 0, nullptr, &s.flag, /*partial_ok=*/false, s.d, s.l, &s.n);
          t_back[t] = std::chrono::steady_clock::now();
          s.status = st.code;
        });
    }
    // the batch of eight is on the device -- on a loaded host a caller can be late for the 20 ms window and travel in a second
    // batch (or wait behind the first): the scenario is not the one under test then, and the caller of this function repeats it
    for (int spin = 0; spin < 1500 && ix.queries.load() < 8; ++spin) std::this_thread::sleep_for(std::chrono::microseconds(100));
    const bool together = ix.calls.load() == 1 && ix.max_batch.load() == 8;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
    t_raise = std::chrono::steady_clock::now();
    __atomic_store_n(const_cast<int *>(&slots[victim]->flag), 1, __ATOMIC_RELAXED);
    if (use_submit) {
      for (int t = 0; t < 8; ++t) {
        while (slots[t]->done.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
        t_back[t] = slots[t]->t_done;
      }
    } else {
      for (auto &t : ts) t.join();
    }
    out[3] = dp.left_early();
    if (!together) return 100;   // (premise not met: see above)
  }
  for (int t = 0; t < 8; ++t) {
    const CancelSlot &s = *slots[t];
    if (t == victim) {
      if (hnsw ? s.status != VK_ERR_CANCELLED : (s.status != VK_OK || s.n != 0)) bad += 1;
    } else {
      if (s.status != VK_OK || s.n != 3 || s.l[0] != (uint64_t)(t + 7) * 10) bad += 1;
      // (the others stayed for the whole pass)
      if (t_back[t] - t0 < std::chrono::milliseconds(150)) bad += 1;
    }
  }
  out[0] = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_back[victim] - t_raise).count();
  out[1] = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  out[2] = ix.member_words_seen;
  return bad.load();
}

// FLAT: `threads` blocking callers, back to back, max_batch = threads.  One pass costs the same for a half-empty batch, so the
// callers must keep travelling TOGETHER (r04: two batches of half the callers each in flight).  out[0] = device passes,
// out[1] = queries served.
extern "C" int dispatcher_flat_fill_run(int threads, int calls, int delay_us, int hnsw, uint64_t *out) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = hnsw ? VK_ALGO_HNSW : VK_ALGO_FLAT;
  FakeIndex ix(p);
  ix.delay_us = delay_us;
  std::atomic<int> bad{0};
  {
    vk::Dispatcher dp(&ix);
    dp.configure((uint32_t)threads, getenv("VK_TEST_WAIT_US") ? (uint32_t)atoi(getenv("VK_TEST_WAIT_US")) : 200);
    dp.set_in_flight(2);
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t)
      ts.emplace_back([&, t] {
        for (int r = 0; r < calls; ++r) {
          const int id = t * calls + r;
          float q[4] = {(float)id, (float)(id + 7), 0.f, 0.f}, d[16];
          uint64_t l[16], n = 99;
          vk::Status st = dp.search(q, 3, 100, nullptr, 0, nullptr, nullptr, true, d, l, &n);
          if (!st.ok() || n != 3 || l[0] != (uint64_t)(id + 7) * 10) bad += 1;
        }
      });
    for (auto &t : ts) t.join();
  }
  out[0] = ix.calls;
  out[1] = ix.queries;
  return bad.load();
}

// A SUBMITTED request whose token goes up while it is still QUEUED behind the batches in flight (one runner, 0.1 s passes, the
// victim three batches back): the watcher answers it within its tick, not when a runner reaches its lane.
// out[0] = microseconds from the token to the callback, out[1] = queries the "device" served (the victim is not among them).
extern "C" int dispatcher_queued_cancel_run(int hnsw, uint64_t *out) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = hnsw ? VK_ALGO_HNSW : VK_ALGO_FLAT;
  FakeIndex ix(p);
  ix.delay_us = 100000;
  std::atomic<int> bad{0};
  const int n = 32, victim = 29;
  std::vector<std::unique_ptr<CancelSlot>> slots;
  for (int t = 0; t < n; ++t) {
    slots.push_back(std::make_unique<CancelSlot>());
    slots[t]->q[0] = (float)t; slots[t]->q[1] = (float)(t + 7); slots[t]->q[2] = slots[t]->q[3] = 0.f;
  }
  std::chrono::steady_clock::time_point t_raise;
  {
    vk::Dispatcher dp(&ix);
    dp.configure(8, 500);
    dp.set_in_flight(1);
    for (int t = 0; t < n; ++t)
      if (!dp.submit(slots[t]->q, 3, 100, nullptr, 0, nullptr, &slots[t]->flag, /*partial_ok=*/false, slots[t]->d, slots[t]->l, &slots[t]->n, cancel_done,
                     slots[t].get()).ok()) bad += 1;
    std::this_thread::sleep_for(std::chrono::milliseconds(30));     // batch 0..7 is on the device, 8..31 wait
    t_raise = std::chrono::steady_clock::now();
    __atomic_store_n(const_cast<int *>(&slots[victim]->flag), 1, __ATOMIC_RELAXED);
    while (slots[victim]->done.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    out[0] = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(slots[victim]->t_done - t_raise).count();
  }   // (the destructor serves the rest)
  for (int t = 0; t < n; ++t) {
    const CancelSlot &s = *slots[t];
    if (t == victim ? (hnsw ? s.status != VK_ERR_CANCELLED : (s.status != VK_OK || s.n != 0)) : (s.status != VK_OK || s.n != 3 || s.l[0] != (uint64_t)(t + 7) * 10)) bad += 1;
  }
  out[1] = ix.queries;
  return bad.load();
}
