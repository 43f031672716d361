// Test helper (not product, not oracle): prints traces of the real libstdc++
// behaviours the oracle restates in oracle/util.c, so tests can pin them.
//   stdlib_trace heap <seed> <ops>   -> pop order of priority_queue<pair<float,unsigned>, vector, CompareByFirst>
//   stdlib_trace rng <seed> <n> <M>  -> minstd_rand0 uniform doubles, hnsw levels, uniform floats
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <random>
#include <vector>

struct CompareByFirst {
  bool operator()(const std::pair<float, unsigned>& a, const std::pair<float, unsigned>& b) const noexcept {
    return a.first < b.first;
  }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  if (!strcmp(argv[1], "heap")) {
    unsigned seed = strtoul(argv[2], nullptr, 10);
    int ops = atoi(argv[3]);
    std::priority_queue<std::pair<float, unsigned>, std::vector<std::pair<float, unsigned>>, CompareByFirst> pq;
    // tiny LCG shared with the python side; few distinct keys => many ties
    unsigned x = seed, id = 0;
    auto next = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (int i = 0; i < ops; ++i) {
      unsigned r = next();
      if ((r % 3) != 0 || pq.empty()) {
        float d = (float)(next() % 7) * 0.5f;
        pq.emplace(d, id++);
      } else {
        printf("%u\n", pq.top().second);
        pq.pop();
      }
    }
    while (!pq.empty()) { printf("%u\n", pq.top().second); pq.pop(); }
    return 0;
  }
  if (!strcmp(argv[1], "rng")) {
    unsigned seed = strtoul(argv[2], nullptr, 10);
    int n = atoi(argv[3]);
    int M = atoi(argv[4]);
    std::default_random_engine g; g.seed(seed);
    std::uniform_real_distribution<double> dd(0.0, 1.0);
    for (int i = 0; i < n; ++i) printf("%.17g\n", dd(g));
    std::default_random_engine g2; g2.seed(seed);
    double mult = 1 / log(1.0 * M);
    for (int i = 0; i < n; ++i) { double r = -log(dd(g2)) * mult; printf("%d\n", (int)r); }
    std::default_random_engine g3; g3.seed(seed + 1);
    std::uniform_real_distribution<float> df(0.0, 1.0);
    for (int i = 0; i < n; ++i) printf("%.9g\n", df(g3));
    return 0;
  }
  return 2;
}
