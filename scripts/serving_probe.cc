// serving_probe.cc -- the path a real FT.SEARCH takes, driven natively against an EXISTING index (no interpreter in the
// loop): single-query requests through the C ABI of include/vk_index.h, either
//   vk_probe_submit    non-blocking (vk_index_search_submit): `producers` threads keep `window` requests outstanding in
//                      all -- the shape of query::SearchAsync (src/query/search.cc:886-910), whose queue holds up to
//                      max-query-queue-depth requests however many reader threads there are, or
//   vk_probe_blocking  `threads` callers of the blocking vk_index_search, back to back (what the reader pool does today).
// Every answer is compared with a reference answer of the same query (ids and distance bits, e.g. from one
// vk_index_search_batch call), latencies are sampled per request.  Built as a shared library (g++, no HIP) by
// __graft_entry__.build(); bench.py and tests/test_submit_gpu.py load it with ctypes and hand it the vk_index* they hold.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "vk_index.h"

extern "C" {
struct vk_probe_result {
  double qps, seconds;
  uint64_t completed, rejected, mismatches, errors;
  uint64_t device_batches, max_batches_in_flight;
  double mean_batch, p50_us, p99_us, max_us;
};
}

namespace {
typedef std::chrono::steady_clock Clock;

struct Producer;
struct Slot {
  Producer *owner;
  uint64_t qi;
  Clock::time_point t0;
  std::vector<float> d;
  std::vector<uint64_t> l;
  uint64_t n = 0;
};
struct Shared {
  const float *ref_d;
  const uint64_t *ref_l;
  uint64_t k;
  std::atomic<uint64_t> completed{0}, mismatches{0}, errors{0};
};
struct Producer {
  Shared *sh;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Slot *> free;
  std::vector<float> lat_us;   // (only touched under mu)
};

bool same_answer(const Shared &sh, const Slot &s) {
  if (!sh.ref_d) return true;
  if (s.n != sh.k) return false;
  return memcmp(s.d.data(), sh.ref_d + s.qi * sh.k, sh.k * 4) == 0 && memcmp(s.l.data(), sh.ref_l + s.qi * sh.k, sh.k * 8) == 0;
}

void on_done(void *user, int status) {
  Slot *s = static_cast<Slot *>(user);
  Producer *p = s->owner;
  const float us = (float)std::chrono::duration<double, std::micro>(Clock::now() - s->t0).count();
  if (status != VK_OK) p->sh->errors.fetch_add(1, std::memory_order_relaxed);
  else if (!same_answer(*p->sh, *s)) p->sh->mismatches.fetch_add(1, std::memory_order_relaxed);
  p->sh->completed.fetch_add(1, std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->lat_us.push_back(us);
    p->free.push_back(s);
  }
  p->cv.notify_one();
}

void percentiles(std::vector<float> &v, vk_probe_result *out) {
  if (v.empty()) return;
  std::sort(v.begin(), v.end());
  out->p50_us = v[v.size() / 2];
  out->p99_us = v[std::min(v.size() - 1, v.size() * 99 / 100)];
  out->max_us = v.back();
}
}  // namespace

extern "C" int vk_probe_submit(vk_index *ix, const float *queries, uint64_t nq, uint32_t dim, uint64_t k, uint64_t ef, int producers,
                               int window, uint64_t total, const float *ref_d, const uint64_t *ref_l, vk_probe_result *out) {
  memset(out, 0, sizeof(*out));
  if (!ix || !queries || nq == 0 || producers < 1 || window < producers) return VK_ERR_INVALID;
  Shared sh;
  sh.ref_d = ref_d;
  sh.ref_l = ref_l;
  sh.k = k;
  std::vector<Producer> ps(producers);
  std::vector<std::vector<Slot>> slots(producers);
  const int per = window / producers;
  for (int p = 0; p < producers; ++p) {
    ps[p].sh = &sh;
    slots[p].resize(per);
    ps[p].lat_us.reserve((size_t)(total / producers + 16));
    for (Slot &s : slots[p]) {
      s.owner = &ps[p];
      s.d.resize(k);
      s.l.resize(k);
      ps[p].free.push_back(&s);
    }
  }
  vk_index_stats st0{}, st1{};
  vk_index_get_stats(ix, &st0);
  std::atomic<uint64_t> rejected{0}, next{0};
  const Clock::time_point t0 = Clock::now();
  std::vector<std::thread> ts;
  for (int p = 0; p < producers; ++p)
    ts.emplace_back([&, p] {
      Producer &me = ps[p];
      for (;;) {
        const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= total) break;
        Slot *s;
        {
          std::unique_lock<std::mutex> lk(me.mu);
          me.cv.wait(lk, [&] { return !me.free.empty(); });
          s = me.free.back();
          me.free.pop_back();
        }
        s->qi = i % nq;
        s->n = 0;
        for (;;) {
          s->t0 = Clock::now();
          const int rc = vk_index_search_submit(ix, queries + s->qi * dim, k, ef, nullptr, 0, nullptr, 1, s->d.data(), s->l.data(), &s->n, on_done, s);
          if (rc == VK_OK) break;
          if (rc != VK_ERR_BUSY) {   // (not queued: the callback will not fire)
            sh.errors.fetch_add(1, std::memory_order_relaxed);
            sh.completed.fetch_add(1, std::memory_order_relaxed);
            std::lock_guard<std::mutex> lk(me.mu);
            me.free.push_back(s);
            break;
          }
          rejected.fetch_add(1, std::memory_order_relaxed);
          std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
      }
      // everything this producer submitted has come back when all its slots are free again
      std::unique_lock<std::mutex> lk(me.mu);
      me.cv.wait(lk, [&] { return me.free.size() == (size_t)per; });
    });
  for (auto &t : ts) t.join();
  out->seconds = std::chrono::duration<double>(Clock::now() - t0).count();
  vk_index_get_stats(ix, &st1);
  out->completed = sh.completed.load();
  out->rejected = rejected.load();
  out->mismatches = sh.mismatches.load();
  out->errors = sh.errors.load();
  out->qps = out->seconds > 0 ? (double)out->completed / out->seconds : 0;
  out->device_batches = st1.coalesced_batches - st0.coalesced_batches;
  out->mean_batch = out->device_batches ? (double)(st1.coalesced_queries - st0.coalesced_queries) / (double)out->device_batches : 0;
  out->max_batches_in_flight = st1.max_batches_in_flight;
  std::vector<float> all;
  for (auto &p : ps) all.insert(all.end(), p.lat_us.begin(), p.lat_us.end());
  percentiles(all, out);
  return VK_OK;
}

extern "C" int vk_probe_blocking(vk_index *ix, const float *queries, uint64_t nq, uint32_t dim, uint64_t k, uint64_t ef, int threads,
                                 int calls_per_thread, const float *ref_d, const uint64_t *ref_l, vk_probe_result *out) {
  memset(out, 0, sizeof(*out));
  if (!ix || !queries || nq == 0 || threads < 1) return VK_ERR_INVALID;
  Shared sh;
  sh.ref_d = ref_d;
  sh.ref_l = ref_l;
  sh.k = k;
  vk_index_stats st0{}, st1{};
  vk_index_get_stats(ix, &st0);
  std::vector<std::vector<float>> lat(threads);
  const Clock::time_point t0 = Clock::now();
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t)
    ts.emplace_back([&, t] {
      Slot s;
      s.d.resize(k);
      s.l.resize(k);
      lat[t].reserve(calls_per_thread);
      for (int r = 0; r < calls_per_thread; ++r) {
        s.qi = ((uint64_t)t * calls_per_thread + r) % nq;
        s.n = 0;
        const Clock::time_point c0 = Clock::now();
        const int rc = vk_index_search(ix, queries + s.qi * dim, k, ef, nullptr, 0, nullptr, 1, s.d.data(), s.l.data(), &s.n);
        lat[t].push_back((float)std::chrono::duration<double, std::micro>(Clock::now() - c0).count());
        if (rc != VK_OK) sh.errors.fetch_add(1, std::memory_order_relaxed);
        else if (!same_answer(sh, s)) sh.mismatches.fetch_add(1, std::memory_order_relaxed);
        sh.completed.fetch_add(1, std::memory_order_relaxed);
      }
    });
  for (auto &t : ts) t.join();
  out->seconds = std::chrono::duration<double>(Clock::now() - t0).count();
  vk_index_get_stats(ix, &st1);
  out->completed = sh.completed.load();
  out->mismatches = sh.mismatches.load();
  out->errors = sh.errors.load();
  out->qps = out->seconds > 0 ? (double)out->completed / out->seconds : 0;
  out->device_batches = st1.coalesced_batches - st0.coalesced_batches;
  out->mean_batch = out->device_batches ? (double)(st1.coalesced_queries - st0.coalesced_queries) / (double)out->device_batches : 0;
  out->max_batches_in_flight = st1.max_batches_in_flight;
  std::vector<float> all;
  for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
  percentiles(all, out);
  return VK_OK;
}

// `threads` writer threads feed n rows ONE AT A TIME through vk_index_add, the way IndexSchema's writer pool feeds AddRecord
// (src/index_schema.cc:755-791; the adaptor's resize-and-retry loop around VK_ERR_CAPACITY included), then the caller flushes.
// Returns the number of failed adds; *seconds = the time of the adds alone.
extern "C" uint64_t vk_probe_add_single(vk_index *ix, const uint64_t *labels, const float *rows, uint64_t n, uint32_t dim, int threads,
                                        uint64_t grow_by, double *seconds) {
  std::atomic<uint64_t> next{0}, failed{0};
  std::mutex resize_mu;
  const Clock::time_point t0 = Clock::now();
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t)
    ts.emplace_back([&] {
      for (;;) {
        const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n) return;
        for (;;) {
          const int rc = vk_index_add(ix, labels ? labels[i] : i, rows + i * dim);
          if (rc == VK_OK) break;
          if (rc != VK_ERR_CAPACITY) { failed.fetch_add(1, std::memory_order_relaxed); break; }
          std::lock_guard<std::mutex> lk(resize_mu);   // (vector_hnsw.cc:238-271 ResizeIfFull)
          vk_index_stats st{};
          vk_index_get_stats(ix, &st);
          if (st.count >= st.capacity && vk_index_resize(ix, st.capacity + (grow_by ? grow_by : 10240)) != VK_OK) { failed.fetch_add(1, std::memory_order_relaxed); break; }
        }
      }
    });
  for (auto &t : ts) t.join();
  if (seconds) *seconds = std::chrono::duration<double>(Clock::now() - t0).count();
  return failed.load();
}
