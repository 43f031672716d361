// N1 without a Python harness in the way: `threads` native callers, each issuing single-query vk_index_search calls
// back to back (what valkey-search's reader pool does, search.cc:886-910), against one FLAT index -- first with
// coalescing off (one device pass per call), then on.  Prints queries/s and the mean device batch.
//   g++ -O2 -std=c++17 -Iinclude scripts/coalescer_native.cc -Lvalkey-search_amd -lvkindex -lpthread \
//       -Wl,-rpath,$PWD/valkey-search_amd -o /tmp/coalescer_native && /tmp/coalescer_native [rows] [dim] [threads] [calls] [hnsw]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "vk_index.h"

static inline uint32_t xs(uint64_t &s) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (uint32_t)(s >> 32);
}
static void unit_rows(float *x, size_t n, size_t dim, uint64_t seed) {
  for (size_t i = 0; i < n; ++i) {
    uint64_t s = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    float *r = x + i * dim;
    double n2 = 0;
    for (size_t j = 0; j < dim; ++j) {
      const uint32_t u = xs(s);
      r[j] = (float)((int)(u & 0xFFFF) + (int)(u >> 16) - 65535) * (1.0f / 65536.0f);   // triangular in (-1, 1)
      n2 += (double)r[j] * r[j];
    }
    const float inv = (float)(1.0 / sqrt(n2));
    for (size_t j = 0; j < dim; ++j) r[j] *= inv;
  }
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000, dim = argc > 2 ? strtoull(argv[2], 0, 10) : 768, k = 10;
  const int threads = argc > 3 ? atoi(argv[3]) : 256, calls = argc > 4 ? atoi(argv[4]) : 100;
  const bool hnsw = argc > 5 && !strcmp(argv[5], "hnsw");
  vk_index_params p{};
  p.struct_size = sizeof p; p.algo = VK_ALGO_FLAT; p.metric = VK_METRIC_COSINE; p.dim = (uint32_t)dim; p.initial_cap = n;
  p.block_size = 1024; p.device_id = 0;
  if (hnsw) { p.algo = VK_ALGO_HNSW; p.m = 16; p.ef_construction = 100; p.ef_runtime = 64; }
  vk_index *ix = nullptr;
  if (vk_index_create(&p, &ix)) { printf("create: %s\n", vk_last_error()); return 1; }
  {
    const size_t chunk = 250000;
    std::vector<float> x(chunk * dim);
    std::vector<uint64_t> lab(chunk);
    for (size_t lo = 0; lo < n; lo += chunk) {
      const size_t c = n - lo < chunk ? n - lo : chunk;
      unit_rows(x.data(), c, dim, 1000 + lo);
      for (size_t i = 0; i < c; ++i) lab[i] = lo + i;
      if (vk_index_add_batch(ix, lab.data(), x.data(), c)) { printf("add: %s\n", vk_last_error()); return 1; }
    }
    vk_index_flush(ix);
  }
  const size_t nq = 4096;
  std::vector<float> q(nq * dim);
  unit_rows(q.data(), nq, dim, 7);
  auto drive = [&](int nt, int per) {
    std::atomic<int> bad{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    for (int t = 0; t < nt; ++t)
      ts.emplace_back([&, t] {
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        uint64_t cnt = 0;
        for (int r = 0; r < per; ++r)
          if (vk_index_search(ix, q.data() + ((size_t)(t * per + r) % nq) * dim, k, 0, nullptr, 0, nullptr, 1, d.data(), l.data(), &cnt) || cnt != k)
            bad += 1;
      });
    for (auto &t : ts) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (bad) printf("  (%d failed calls)\n", bad.load());
    return (double)nt * per / s;
  };
  vk_index_stats st0{}, st1{};
  vk_index_set_coalescing(ix, 0, 0);
  drive(8, 4);
  const double q0 = drive(threads < 32 ? threads : 32, 8);
  printf("%s %zux%zu cosine k=%zu, single-query calls from native threads\n", hnsw ? "HNSW (M=16, ef=64)" : "FLAT", n, dim, k);
  printf("  coalescing off, %d callers: %.0f queries/s (one device pass per call)\n", threads < 32 ? threads : 32, q0);
  for (uint32_t wait_us : {100u, 300u, 1000u}) {
    vk_index_set_coalescing(ix, 256, wait_us);
    drive(threads, 4);
    vk_index_get_stats(ix, &st0);
    const double q1 = drive(threads, calls);
    vk_index_get_stats(ix, &st1);
    const double nb = (double)(st1.coalesced_batches - st0.coalesced_batches);
    printf("  coalescing on (max_batch 256, max_wait %u us), %d callers x %d calls: %.0f queries/s, %.0f device batches, mean batch %.1f\n",
           wait_us, threads, calls, q1, nb, nb > 0 ? (double)threads * calls / nb : 0.0);
  }
  vk_index_set_coalescing(ix, 0, 0);
  vk_index_destroy(ix);
  return 0;
}
