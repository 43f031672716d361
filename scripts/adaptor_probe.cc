// adaptor_probe.cc -- single-query FT.SEARCH traffic THROUGH THE ADAPTOR CLASSES (include/vk_vector_adaptor.h: VectorGpuFlat /
// VectorGpuHNSW derived from the mock of VectorBase, tests/helpers/mock_valkey_search.h), driven natively against an index
// the benchmark already holds (adopted, not owned).  The shape is query::SearchAsync's (src/query/search.cc:886-910):
//   `fronts` front threads (the main thread and the io threads that parse commands next to it) keep `window` FT.SEARCH
//   requests outstanding between them (the clients' concurrency) and schedule each on a reader pool of `readers` threads
//   (reader-threads = the box's cores); the pool task calls
//     async    VectorGpu::SearchAsync -- returns after the submission; the library's completion thread takes the reply
//              (CreateReply's key lookups already done), compares it and hands the request slot back to its front thread
//              (the RunByMain step of the reference's callback), or
//     blocking VectorGpu::Search      -- what PerformVectorSearch does today (search.cc:135-170): the pool thread is parked
//              until the answer is there, so at most `readers` queries are in flight.
// Every reply is compared with a reference answer of the same query (ids and distance bits).  bench.py:
// single_query_serving.adaptor.  Built by __graft_entry__.build() as a shared library (g++, no HIP).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "mock_valkey_search.h"
#include "vk_vector_adaptor.h"

extern "C" int ValkeyModule_ReplyWithSimpleString(ValkeyModuleCtx *, const char *) { return 0; }
extern "C" int ValkeyModule_ReplyWithLongLong(ValkeyModuleCtx *, long long) { return 0; }

extern "C" {
struct vk_probe_result {   // == scripts/serving_probe.cc
  double qps, seconds;
  uint64_t completed, rejected, mismatches, errors;
  uint64_t device_batches, max_batches_in_flight;
  double mean_batch, p50_us, p99_us, max_us;
};
}

namespace {
using namespace valkey_search;
using namespace valkey_search::indexes;
typedef std::chrono::steady_clock Clock;

struct NeverCancelled : cancel::Base {
  bool IsCancelled() override { return false; }
  void Cancel() override {}
};

// vmsdk::ThreadPool's role -- `n` workers taking tasks in order -- with a queue PER WORKER (tasks are dealt round robin): with
// one queue, one mutex, the 4 lock operations per request of 16 workers, the front threads and the library's completer
// threads were what bounded the request rate (r05_pipeline1.log: 15-60 us inside each completion callback)
class Pool {
 public:
  explicit Pool(int n) : ws_((size_t)n) {
    for (int i = 0; i < n; ++i) ws_[(size_t)i].t = std::thread([this, i] { Run(ws_[(size_t)i]); });
  }
  ~Pool() {
    for (auto &w : ws_) {
      { std::lock_guard<std::mutex> lk(w.mu); w.stop = true; }
      w.cv.notify_all();
    }
    for (auto &w : ws_) w.t.join();
  }
  void Schedule(std::function<void()> f) {
    Worker &w = ws_[next_.fetch_add(1, std::memory_order_relaxed) % ws_.size()];
    bool wake;
    { std::lock_guard<std::mutex> lk(w.mu); w.q.push_back(std::move(f)); wake = w.idle; }
    if (wake) w.cv.notify_one();
  }

 private:
  struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::thread t;
    bool idle = false, stop = false;
  };
  void Run(Worker &w) {
    std::deque<std::function<void()>> mine;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(w.mu);
        w.idle = true;
        w.cv.wait(lk, [&] { return w.stop || !w.q.empty(); });
        w.idle = false;
        if (w.q.empty()) return;
        mine.swap(w.q);   // (everything that is there: one lock operation for the lot)
      }
      for (auto &f : mine) f();
      mine.clear();
    }
  }
  std::vector<Worker> ws_;
  std::atomic<uint64_t> next_{0};
};

struct Run {
  const float *ref_d;
  const uint64_t *ref_l;
  uint64_t k;
  std::atomic<uint64_t> completed{0}, mismatches{0}, errors{0}, rejected{0};
  bool same(uint64_t qi, const std::vector<Neighbor> &r) const {
    if (!ref_d) return true;
    if (r.size() != k) return false;
    for (uint64_t i = 0; i < k; ++i) {
      if (memcmp(&r[i].distance, ref_d + qi * k + i, 4) != 0) return false;
      uint64_t id = 0;   // (the mock's key of internal id N is the string "N")
      for (char c : r[i].external_id->Str()) id = id * 10 + (uint64_t)(c - '0');
      if (id != ref_l[qi * k + i]) return false;
    }
    return true;
  }
};

// one front thread (the main thread, or one of the io threads that parse commands next to it): its share of the clients'
// concurrency and of the requests
struct Front;
// what a completer thread has answered for one front thread inside the span it is handing over (the bulk sink below): the
// replies of a piece go back to their front thread in ONE queue operation and one wake -- the RunByMain step, once per span
struct Pending {
  Front *front;
  std::vector<float> lat_us;
  uint64_t errors = 0, mismatches = 0;
};
thread_local std::vector<Pending> *t_span = nullptr;   // non-null while the bulk sink runs on this thread

struct Front {
  Run *run = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  int free_slots = 0, window = 0;
  bool waiting = false;
  std::vector<float> lat_us;
  void finish(uint64_t qi, Clock::time_point t0, const absl::StatusOr<std::vector<Neighbor>> &r) {
    const float us = (float)std::chrono::duration<double, std::micro>(Clock::now() - t0).count();
    const bool err = !r.ok(), mis = !err && !run->same(qi, r.value());
    if (t_span) {   // inside a span: collected, handed over by flush()
      Pending *p = nullptr;
      for (Pending &x : *t_span)
        if (x.front == this) p = &x;
      if (!p) { t_span->push_back(Pending{this, {}, 0, 0}); p = &t_span->back(); }
      p->lat_us.push_back(us);
      p->errors += err;
      p->mismatches += mis;
      return;
    }
    if (err) run->errors.fetch_add(1, std::memory_order_relaxed);
    if (mis) run->mismatches.fetch_add(1, std::memory_order_relaxed);
    run->completed.fetch_add(1, std::memory_order_relaxed);
    bool wake;
    {
      std::lock_guard<std::mutex> lk(mu);
      lat_us.push_back(us);
      free_slots += 1;
      wake = waiting;
    }
    if (wake) cv.notify_one();
  }
  void flush(Pending &p) {
    if (p.errors) run->errors.fetch_add(p.errors, std::memory_order_relaxed);
    if (p.mismatches) run->mismatches.fetch_add(p.mismatches, std::memory_order_relaxed);
    run->completed.fetch_add(p.lat_us.size(), std::memory_order_relaxed);
    bool wake;
    {
      std::lock_guard<std::mutex> lk(mu);
      lat_us.insert(lat_us.end(), p.lat_us.begin(), p.lat_us.end());
      free_slots += (int)p.lat_us.size();
      wake = waiting;
    }
    if (wake) cv.notify_one();
  }
};

template <class Ix>
int drive(Ix &ix, const float *queries, uint64_t nq, uint32_t dim, uint64_t k, uint64_t ef, int readers, int fronts, int window, uint64_t total,
          int blocking, Run &run, vk_probe_result *out) {
  vk_index_stats st0{}, st1{};
  vk_index_get_stats(ix.handle(), &st0);
  fronts = std::max(1, std::min(fronts, window));
  std::vector<std::unique_ptr<Front>> fs;
  for (int f = 0; f < fronts; ++f) {
    fs.emplace_back(new Front());
    fs.back()->run = &run;
    fs.back()->window = fs.back()->free_slots = window / fronts + (f < window % fronts ? 1 : 0);
    fs.back()->lat_us.reserve((size_t)(total / fronts) + 16);
  }
  cancel::Token token = std::make_shared<NeverCancelled>();
  // the replies of a piece of a batch arrive together (VectorGpu::SetBulkDone): each is checked, then every front thread
  // gets ITS share in one go
  // (A/B inside one process: VK_PROBE_BULK=0 takes the hook off again -- one callback per request, as r05 had it)
  const bool bulk = !getenv("VK_PROBE_BULK") || atoi(getenv("VK_PROBE_BULK")) != 0;
  if (!bulk) vk_index_set_batch_completion(ix.handle(), nullptr, nullptr);
  if (!blocking && bulk)
    ix.SetBulkDone([](std::vector<typename Ix::CompletedSearch> &&span) {
      std::vector<Pending> mine;
      t_span = &mine;
      for (auto &c : span) c.done(std::move(c.result));
      t_span = nullptr;
      for (Pending &p : mine) p.front->flush(p);
    });
  const Clock::time_point t0 = Clock::now();
  {
    Pool pool(readers);
    auto front_loop = [&](int f) {
      Front &me = *fs[f];
      const uint64_t share = total / fronts + ((uint64_t)f < total % fronts ? 1 : 0);
      int slots = 0;
      for (uint64_t i = 0; i < share; ++i) {
        if (slots == 0) {   // the clients' concurrency: `window` requests outstanding (every free slot is taken at once)
          std::unique_lock<std::mutex> lk(me.mu);
          me.waiting = true;
          me.cv.wait(lk, [&] { return me.free_slots > 0; });
          me.waiting = false;
          slots = me.free_slots;
          me.free_slots = 0;
        }
        slots -= 1;
        const uint64_t qi = (i * fronts + f) % nq;
        const Clock::time_point ts = Clock::now();
        pool.Schedule([&, qi, ts] {
          absl::string_view q(reinterpret_cast<const char *>(queries + qi * dim), (size_t)dim * 4);
          std::optional<size_t> efo;
          if (ef) efo = (size_t)ef;
          if (blocking) {
            cancel::Token tk = token;
            me.finish(qi, ts, ix.Search(q, k, tk, VkFilterRef(), efo, false));
            return;
          }
          for (;;) {
            absl::Status st = ix.SearchAsync(q, k, token, VkFilterRef(), efo, false, [&me, qi, ts](absl::StatusOr<std::vector<Neighbor>> r) {
              // (a library thread in the role of the pool thread that runs the callback of query::SearchAsync today,
              //  search.cc:905-908: the neighbours are checked here and the slot goes back to the front thread -- RunByMain)
              me.finish(qi, ts, r);
            });
            if (st.ok()) break;
            if (st.code() != absl::StatusCode::kResourceExhausted) { me.finish(qi, ts, st); break; }
            run.rejected.fetch_add(1, std::memory_order_relaxed);
            std::this_thread::sleep_for(std::chrono::microseconds(50));
          }
        });
      }
      std::unique_lock<std::mutex> lk(me.mu);
      me.free_slots += slots;
      me.waiting = true;
      me.cv.wait(lk, [&] { return me.free_slots == me.window; });
    };
    std::vector<std::thread> ts;
    for (int f = 1; f < fronts; ++f) ts.emplace_back(front_loop, f);
    front_loop(0);
    for (auto &t : ts) t.join();
  }   // (the pool drains and joins)
  out->seconds = std::chrono::duration<double>(Clock::now() - t0).count();
  vk_index_get_stats(ix.handle(), &st1);
  out->completed = run.completed.load();
  out->rejected = run.rejected.load();
  out->mismatches = run.mismatches.load();
  out->errors = run.errors.load();
  out->qps = out->seconds > 0 ? (double)out->completed / out->seconds : 0;
  out->device_batches = st1.coalesced_batches - st0.coalesced_batches;
  out->mean_batch = out->device_batches ? (double)(st1.coalesced_queries - st0.coalesced_queries) / (double)out->device_batches : 0;
  out->max_batches_in_flight = st1.max_batches_in_flight;
  std::vector<float> v;
  for (auto &f : fs) v.insert(v.end(), f->lat_us.begin(), f->lat_us.end());
  if (!v.empty()) {
    std::sort(v.begin(), v.end());
    out->p50_us = v[v.size() / 2];
    out->p99_us = v[std::min(v.size() - 1, v.size() * 99 / 100)];
    out->max_us = v.back();
  }
  return VK_OK;
}
}  // namespace

// `ix` stays the caller's; max_batch / wait_us: the coalescing the adaptor would set from reader-threads (0 = its defaults)
extern "C" int vk_adaptor_probe(vk_index *ix, int hnsw, uint32_t dim, uint32_t m, const float *queries, uint64_t nq, uint64_t k, uint64_t ef,
                                int readers, int fronts, int window, uint64_t total, int blocking, uint32_t max_batch, uint32_t wait_us, const float *ref_d,
                                const uint64_t *ref_l, vk_probe_result *out) {
  memset(out, 0, sizeof(*out));
  if (!ix || !queries || nq == 0 || readers < 1 || window < 1) return VK_ERR_INVALID;
  data_model::VectorIndex proto;
  proto.dimension_count_ = dim;
  // (rows and queries are normalised by the benchmark already: the inner-product space, no second normalisation of the query)
  proto.distance_metric_ = data_model::DISTANCE_METRIC_IP;
  proto.hnsw_.m_ = m ? m : 16;
  Run run;
  run.ref_d = ref_d;
  run.ref_l = ref_l;
  run.k = k;
  int rc;
  if (hnsw) {
    auto a = VectorGpuHNSW<float>::FromHandle(ix, proto, "v", data_model::ATTRIBUTE_DATA_TYPE_HASH, (uint32_t)readers);
    if (!a.ok()) return VK_ERR_INTERNAL;
    a.value()->MockAllKeysLive();
    if (max_batch) vk_index_set_coalescing(ix, max_batch, wait_us);
    rc = drive(*a.value(), queries, nq, dim, k, ef, readers, fronts, window, total, blocking, run, out);
  } else {
    auto a = VectorGpuFlat<float>::FromHandle(ix, proto, "v", data_model::ATTRIBUTE_DATA_TYPE_HASH, (uint32_t)readers);
    if (!a.ok()) return VK_ERR_INTERNAL;
    a.value()->MockAllKeysLive();
    if (max_batch) vk_index_set_coalescing(ix, max_batch, wait_us);
    rc = drive(*a.value(), queries, nq, dim, k, ef, readers, fronts, window, total, blocking, run, out);
  }
  return rc;
}
