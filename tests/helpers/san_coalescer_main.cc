// Sanitizer harness (ASAN and TSAN builds, tests/test_host_sanitizers.py) for csrc/dispatcher.hpp: the scenarios of
// coalescer_shim.cc as an executable, plus the one ADVICE r02 found by reading -- a FILTERED follower whose
// cancellation token goes up while the batch it was popped into is on the device.  The module frees the query and the
// bitmap as soon as vk_index_search returns (the reference's SearchParameters own both, src/query/search.h), so the
// request may leave early only while it is still queued; here the caller frees both the moment the call returns and
// the fake index reads the bitmap AFTER its "device pass" -- a heap-use-after-free under ASAN with the old code.
#include <cstdlib>
#include <memory>

#include "coalescer_shim.cc"

static int cancel_mid_batch(int rounds) {
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  p.algo = VK_ALGO_HNSW;
  FakeIndex ix(p);
  ix.delay_us = 3000;                       // a long device pass: the followers' flags go up in the middle of it
  vk::Dispatcher co(&ix);
  co.configure(8, 20000);
  std::atomic<int> bad{0};
  for (int r = 0; r < rounds; ++r) {
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; ++t) {
      ts.emplace_back([&, t] {
        const int id = r * 8 + t;
        auto q = std::make_unique<float[]>(4);
        q[0] = (float)id; q[1] = (float)(id + 7); q[2] = q[3] = 0.f;
        auto bits = std::make_unique<uint64_t[]>(1);
        bits[0] = (uint64_t)id * 1000;
        volatile int flag = 0;
        float d[16];
        uint64_t l[16], n = 99;
        std::thread raiser;
        const bool cancels = t % 2 == 1;
        if (cancels) raiser = std::thread([&flag] { std::this_thread::sleep_for(std::chrono::microseconds(1200)); __atomic_store_n(const_cast<int *>(&flag), 1, __ATOMIC_RELAXED); });
        // rounds alternate: a raw host bitmap pins the caller to its batch (the runner reads it); without one the caller
        // LEAVES when its token goes up -- its query, its token and its output buffers die while the batch is on the device
        const bool pinned = r % 2 == 0;
        vk::Status st = co.search(q.get(), 3, 100, pinned ? bits.get() : nullptr, pinned ? 64 : 0, nullptr, &flag, /*partial_ok=*/t % 4 == 1, d, l, &n);
        q.reset();                           // the module's buffers die with the call
        bits.reset();
        if (raiser.joinable()) raiser.join();
        if (!cancels) {
          if (!st.ok() || n != 3 || l[0] != (uint64_t)(id + 7) * 10 + (pinned ? (uint64_t)id * 1000 + 64 : 0)) bad += 1;
        } else if (t % 4 == 1) {             // partial results wanted: an answer or nothing, never an error
          if (!st.ok()) bad += 1;
        } else {                             // HNSW without partial results: cancelled (or served before the flag)
          if (!st.ok() && st.code != VK_ERR_CANCELLED) bad += 1;
        }
      });
    }
    for (auto &t : ts) t.join();
  }
  return bad.load();
}

int main(int argc, char **argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 1;
  uint64_t out[8];
  int bad = 0;
  bad += coalescer_run(24 * scale, 20, 16, 2000, 1, 0, out);
  bad += coalescer_run(16 * scale, 20, 8, 1000, 3, 0, out);
  bad += coalescer_run(12 * scale, 16, 8, 1000, 4, 1, out);
  bad += coalescer_run(16 * scale, 20, 16, 2000, 1, 2, out);
  bad += cancel_mid_batch(6 * scale);
  // the non-blocking entry: submit / completion callbacks / VK_ERR_BUSY / destroy with work in flight / the batch's own token
  bad += dispatcher_async_run(6 * scale, 300, 64, 32, 500, 2, 100000, 300, 0, out);
  bad += dispatcher_async_run(4 * scale, 200, 200, 16, 500, 3, 40, 300, 1, out);   // a shallow queue: rejections
  bad += dispatcher_async_run(8, 1200 * scale, 1200, 4096, 3000, 2, 100000, 2000, 1, out);   // batches beyond 1024: answered by the completer threads
  bad += dispatcher_destroy_run(6 * scale, 50);
  bad += dispatcher_two_indexes_run(4 * scale, 600);
  bad += dispatcher_batch_cancel_run(1, out);
  bad += dispatcher_batch_cancel_run(0, out);
  // one member of a live batch cancelled: blocking and submitted, both index kinds
  for (int hnsw = 0; hnsw < 2; ++hnsw)
    for (int sub = 0; sub < 2; ++sub) {
      // (100 = the eight callers did not make ONE batch -- under a sanitizer threads start slowly: the scenario is repeated;
      //  what is checked here is the sanitizer's verdict on whatever interleaving happened, not the scenario's timing)
      int rc = 100;
      for (int attempt = 0; attempt < 4 && rc == 100; ++attempt) rc = dispatcher_member_cancel_run(hnsw, sub, 3, out);
      bad += rc == 100 ? 0 : rc;
    }
  bad += dispatcher_flat_fill_run(32, 6, 2000, 0, out);
  bad += dispatcher_queued_cancel_run(1, out);
  bad += dispatcher_queued_cancel_run(0, out);
  printf("bad=%d\n", bad);
  return bad ? 1 : 0;
}
