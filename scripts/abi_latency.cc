// Single-query latency through the C ABI without any binding overhead: FLAT 100k x 128 L2 k=10 (config 1).
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "vk_index.h"
int main() {
  const size_t n = 100000, dim = 128, k = 10;
  std::mt19937 g(1);
  std::normal_distribution<float> nd;
  std::vector<float> x(n * dim), q(dim * 64);
  for (auto &v : x) v = nd(g);
  for (auto &v : q) v = nd(g);
  vk_index_params p{};
  p.struct_size = sizeof p; p.algo = VK_ALGO_FLAT; p.metric = VK_METRIC_L2; p.dim = dim; p.initial_cap = n; p.block_size = 1024; p.device_id = -1;
  vk_index *ix = nullptr;
  if (vk_index_create(&p, &ix)) { printf("create: %s\n", vk_last_error()); return 1; }
  vk_index_add_batch(ix, nullptr, x.data(), n);
  vk_index_flush(ix);
  std::vector<float> d(k); std::vector<uint64_t> l(k); uint64_t cnt = 0;
  for (int i = 0; i < 20; ++i) vk_index_search(ix, q.data(), k, 0, nullptr, 0, nullptr, 1, d.data(), l.data(), &cnt);
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 2000;
  for (int i = 0; i < reps; ++i) vk_index_search(ix, q.data() + (i % 64) * dim, k, 0, nullptr, 0, nullptr, 1, d.data(), l.data(), &cnt);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("vk_index_search FLAT %zux%zu L2 k=%zu: %.1f us per call (host in, host out)\n", n, dim, k, us);
  vk_index_destroy(ix);
  return 0;
}
