// Compile-and-run check of include/vk_algo.h: the calls VectorFlat<float> / VectorHNSW<float> make on
// their `algo_` (Create, addPoint in a resize-and-retry loop, searchKnn with ef / filter / cancel flag,
// markDelete / removePoint, distance, SaveIndex -> LoadIndex), for both algorithms.  Prints one line per
// check; the Python test compares the search lines with the ctypes binding's answers on the same data.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vk_algo.h"

using Algo = vkalgo::Algo<float>;

static std::vector<float> read_f32(const char *path, size_t count) {
  std::vector<float> x(count);
  FILE *f = fopen(path, "rb");
  if (!f || fread(x.data(), 4, count, f) != count) throw std::runtime_error(std::string("cannot read ") + path);
  fclose(f);
  return x;
}

static void print_result(const char *tag, Algo::ResultQueue q) {
  std::vector<std::pair<float, size_t>> v;
  while (!q.empty()) { v.push_back(q.top()); q.pop(); }
  printf("%s", tag);
  for (auto it = v.rbegin(); it != v.rend(); ++it) {
    uint32_t bits;
    memcpy(&bits, &it->first, 4);
    printf(" %zu:%08x", it->second, bits);
  }
  printf("\n");
}

struct Chunks { std::vector<std::string> c; size_t next = 0; };
static int write_chunk(void *u, const void *d, uint64_t n) {
  static_cast<Chunks *>(u)->c.emplace_back(static_cast<const char *>(d), n);
  return 0;
}
static int read_chunk(void *u, void *buf, uint64_t cap, uint64_t *len) {
  Chunks *c = static_cast<Chunks *>(u);
  if (c->next >= c->c.size()) return 1;
  const std::string &s = c->c[c->next++];
  if (s.size() > cap) return 2;
  memcpy(buf, s.data(), s.size());
  *len = s.size();
  return 0;
}

static int run(vk_algo algo, const char *name, const std::vector<float> &x, const std::vector<float> &q) {
  const size_t n = 3000, dim = 24, k = 5;
  // Create with a small capacity: AddRecordImpl's loop catches "exceeds the specified limit", resizes by
  // block_size and retries (vector_flat.cc:165-176, vector_hnsw.cc:186-197)
  Algo a(algo, (uint32_t)dim, VK_METRIC_L2, /*max_elements=*/1000, 16, 100, /*block_size=*/1024);
  size_t resizes = 0;
  for (size_t i = 0; i < n; ++i) {
    for (;;) {
      try {
        a.addPoint(x.data() + i * dim, i);
        break;
      } catch (const std::exception &e) {
        if (std::string(e.what()).find("exceeds the specified limit") == std::string::npos) throw;
        a.resizeIndex(a.getMaxElements() + 1024);
        ++resizes;
      }
    }
  }
  printf("%s count %zu capacity %zu resizes %zu\n", name, a.getCurrentElementCount(), a.getMaxElements(), resizes);
  a.flush();
  for (int i = 0; i < 4; ++i) print_result((std::string(name) + " knn" + std::to_string(i)).c_str(), a.searchKnn(q.data() + i * dim, k, 64));
  // filter as an allow-bitmap over labels: even labels only
  std::vector<uint64_t> bits((n + 63) / 64, 0);
  for (size_t l = 0; l < n; l += 2) bits[l >> 6] |= 1ull << (l & 63);
  print_result((std::string(name) + " even").c_str(), a.searchKnn(q.data(), k, 64, bits.data(), n));
  // a cancelled search that may not return partial results throws the reference's timeout message
  volatile int cancel = 1;
  try {
    a.searchKnn(q.data(), k, 64, nullptr, 0, &cancel, /*partial_ok=*/false);
    printf("%s cancel no-throw\n", name);
  } catch (const vkalgo::Error &e) {
    printf("%s cancel code %d msg %s\n", name, e.code, e.what());
  }
  // deletes
  if (algo == VK_ALGO_FLAT) a.removePoint(0); else a.markDelete(0);
  printf("%s after-delete count %zu deleted %zu\n", name, a.getCurrentElementCount(), a.getDeletedCount());
  print_result((std::string(name) + " self1").c_str(), a.searchKnn(x.data() + dim, 1, 64));
  printf("%s dist(7,q0) %.9g\n", name, (double)a.distance(7, q.data()));
  const uint64_t some[6] = {5, 9, 11, 2999, 1234, 77};
  print_result((std::string(name) + " prefilter").c_str(), a.searchLabels(q.data(), 3, some, 6));
  // SaveIndex -> LoadIndex
  Chunks ch;
  a.SaveIndex(write_chunk, &ch);
  vk_index_params p = a.params();
  p.initial_cap = a.getMaxElements();
  Algo b(p, read_chunk, &ch);
  printf("%s reloaded count %zu chunks %zu\n", name, b.getCurrentElementCount(), ch.c.size());
  print_result((std::string(name) + " knn0-reloaded").c_str(), b.searchKnn(q.data(), k, 64));
  return 0;
}

int main(int argc, char **argv) {
  try {
    if (argc < 3) throw std::runtime_error("usage: vk_algo_check rows.f32 queries.f32   (3000x24 and 4x24)");
    const std::vector<float> x = read_f32(argv[1], 3000 * 24), q = read_f32(argv[2], 4 * 24);
    run(VK_ALGO_FLAT, "FLAT", x, q);
    run(VK_ALGO_HNSW, "HNSW", x, q);
  } catch (const std::exception &e) {
    printf("FAILED %s\n", e.what());
    return 1;
  }
  return 0;
}
