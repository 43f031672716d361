// Compile-and-run check of include/vk_vector_adaptor.h -- the VectorBase-derived binding -- against the MOCK of the
// module's interface (tests/helpers/mock_valkey_search.h).  `-Wall -Werror` on every CPU run (tests/test_abi_symbols.py:
// the overrides must match the virtuals' signatures); on a GPU (tests/test_facade_gpu.py) it is run: AddRecord through the
// resize-and-retry loop, Search with ef / filter functor / cancellation token / partial results, Modify / Remove,
// ComputeDistanceFromRecord, GetValue, ToProto, RespondWithInfo, SaveIndex -> vk_index_load.  One line per check; the
// Python test compares the search lines with the ctypes binding's answers on the same data.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mock_valkey_search.h"
#include "vk_vector_adaptor.h"

using namespace valkey_search;
using namespace valkey_search::indexes;

static std::vector<std::string> g_info;
extern "C" int ValkeyModule_ReplyWithSimpleString(ValkeyModuleCtx *, const char *msg) { g_info.emplace_back(msg); return 0; }
extern "C" int ValkeyModule_ReplyWithLongLong(ValkeyModuleCtx *, long long v) { g_info.emplace_back(std::to_string(v)); return 0; }

static std::vector<float> read_f32(const char *path, size_t count) {
  std::vector<float> x(count);
  FILE *f = fopen(path, "rb");
  if (!f || fread(x.data(), 4, count, f) != count) throw std::runtime_error(std::string("cannot read ") + path);
  fclose(f);
  return x;
}

struct EvenOnly : hnswlib::BaseFilterFunctor {
  bool operator()(hnswlib::labeltype id) override { return id % 2 == 0; }
};
struct Token : cancel::Base {
  bool c = false;
  bool IsCancelled() override { return c; }
  void Cancel() override { c = true; }
};
struct Chunks : RDBChunkOutputStream {
  std::vector<std::string> c;
  size_t next = 0;
  absl::Status SaveChunk(const char *d, size_t n) override { c.emplace_back(d, n); return absl::OkStatus(); }
};
static int read_chunk(void *u, void *buf, uint64_t cap, uint64_t *len) {
  Chunks *c = static_cast<Chunks *>(u);
  if (c->next >= c->c.size()) return 1;
  const std::string &s = c->c[c->next++];
  if (s.size() > cap) return 2;
  memcpy(buf, s.data(), s.size());
  *len = s.size();
  return 0;
}

// what Tag::Search / Numeric::Search hand to the query layer (index_base.h:103-116): a list of keys
struct ListFetcher : EntriesFetcherBase {
  std::vector<InternedStringPtr> keys;
  struct It : EntriesFetcherIteratorBase {
    const std::vector<InternedStringPtr> *k;
    size_t i = 0;
    explicit It(const std::vector<InternedStringPtr> *keys) : k(keys) {}
    bool Done() const override { return i >= k->size(); }
    void Next() override { ++i; }
    const InternedStringPtr &operator*() const override { return (*k)[i]; }
  };
  size_t Size() const override { return keys.size(); }
  std::unique_ptr<EntriesFetcherIteratorBase> Begin() override { return std::make_unique<It>(&keys); }
};
static std::unique_ptr<EntriesFetcherBase> fetcher_of(size_t n, size_t step, bool with_strangers) {
  auto f = std::make_unique<ListFetcher>();
  for (size_t i = 0; i < n; i += step) f->keys.push_back(std::make_shared<InternedString>(std::to_string(i)));
  if (with_strangers)   // keys of the schema that have no vector in this index: not candidates
    for (size_t i = 0; i < 5; ++i) f->keys.push_back(std::make_shared<InternedString>("doc:" + std::to_string(i)));
  return f;
}
// one asynchronous search, waited for here (the module re-posts the completion to its pool instead)
struct Latch {
  std::atomic<int> done{0};
  absl::StatusOr<std::vector<Neighbor>> result{absl::InternalError("not completed")};
  void wait() const { while (!done.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50)); }
};

static void print_result(const char *tag, const std::vector<Neighbor> &v) {
  printf("%s", tag);
  for (const Neighbor &nb : v) {
    uint32_t bits;
    memcpy(&bits, &nb.distance, 4);
    printf(" %s:%08x", std::string(nb.external_id->Str()).c_str(), bits);
  }
  printf("\n");
}

template <class Ix>
static int run(Ix &ix, const char *name, const std::vector<float> &x, const std::vector<float> &q, size_t n, size_t dim,
               const data_model::VectorIndex &definition) {
  const size_t k = 5;
  const size_t cap0 = ix.GetCapacity();
  for (size_t i = 0; i < n; ++i) {
    absl::string_view rec(reinterpret_cast<const char *>(x.data() + i * dim), dim * 4);
    auto key = std::make_shared<InternedString>(std::string(rec));
    if (!ix.MockAdd(i, rec, key).ok()) { printf("%s add failed at %zu\n", name, i); return 1; }
  }
  printf("%s capacity %zu -> %zu count %zu max_label %llu\n", name, cap0, ix.GetCapacity(), ix.GetLabelCount(),
         (unsigned long long)ix.GetMaxInternalLabel());
  cancel::Token tok = std::make_shared<Token>();
  for (size_t i = 0; i < 3; ++i) {
    absl::string_view qs(reinterpret_cast<const char *>(q.data() + i * dim), dim * 4);
    auto r = ix.Search(qs, k, tok, nullptr, 64);
    if (!r.ok()) { printf("%s search failed: %s\n", name, r.status().message().c_str()); return 1; }
    print_result((std::string(name) + " q" + std::to_string(i)).c_str(), r.value());
    auto rf = ix.Search(qs, k, tok, std::make_unique<EvenOnly>(), 64);
    if (!rf.ok()) return 1;
    print_result((std::string(name) + " q" + std::to_string(i) + " even").c_str(), rf.value());
  }
  // ---- SearchAsync: the same answers without a parked thread; 256 of them in flight at once
  for (size_t i = 0; i < 3; ++i) {
    absl::string_view qs(reinterpret_cast<const char *>(q.data() + i * dim), dim * 4);
    Latch l;
    auto st = ix.SearchAsync(qs, k, tok, VkFilterRef(), 64, false, [&l](absl::StatusOr<std::vector<Neighbor>> r) { l.result = std::move(r); l.done.store(1, std::memory_order_release); });
    if (!st.ok()) { printf("%s async submit failed: %s\n", name, st.message().c_str()); return 1; }
    l.wait();
    if (!l.result.ok()) return 1;
    print_result((std::string(name) + " async q" + std::to_string(i)).c_str(), l.result.value());
  }
  {
    const int kBurst = 256;
    std::vector<std::unique_ptr<Latch>> ls;
    for (int i = 0; i < kBurst; ++i) ls.push_back(std::make_unique<Latch>());
    absl::string_view q0s(reinterpret_cast<const char *>(q.data()), dim * 4);
    for (int i = 0; i < kBurst; ++i) {
      Latch *l = ls[(size_t)i].get();
      if (!ix.SearchAsync(q0s, k, tok, VkFilterRef(), 64, false, [l](absl::StatusOr<std::vector<Neighbor>> r) { l->result = std::move(r); l->done.store(1, std::memory_order_release); }).ok()) return 1;
    }
    int same = 0;
    for (auto &l : ls) {
      l->wait();
      if (!l->result.ok()) return 1;
      const auto &a = l->result.value(), &b = ls[0]->result.value();
      bool eq = a.size() == b.size();
      for (size_t j = 0; eq && j < a.size(); ++j) eq = a[j].external_id->Str() == b[j].external_id->Str() && a[j].distance == b[j].distance;
      same += eq;
    }
    print_result((std::string(name) + " burst q0").c_str(), ls[0]->result.value());
    printf("%s burst %d in flight, %d identical\n", name, kBurst, same);
  }
  // ---- filters from the EntriesFetchers of a predicate: the union of two overlapping key lists (plus keys this index does
  // not hold), then a list that needs the per-key predicate; cached under the predicate's text until the next write phase
  {
    vk_index_stats s0, s1, s2, s3;
    vk_index_get_stats(ix.handle(), &s0);
    std::queue<std::unique_ptr<EntriesFetcherBase>> fq;
    fq.push(fetcher_of(n, 2, true));
    fq.push(fetcher_of(n, 6, false));
    auto f1 = ix.BuildFilter(fq, nullptr, "@parity:{even}");
    if (!f1.ok()) { printf("%s BuildFilter failed: %s\n", name, f1.status().message().c_str()); return 1; }
    vk_index_get_stats(ix.handle(), &s1);
    std::queue<std::unique_ptr<EntriesFetcherBase>> fq2;
    fq2.push(fetcher_of(n, 2, true));
    auto f2 = ix.BuildFilter(fq2, nullptr, "@parity:{even}");          // served from the cache: nothing is walked or built
    if (!f2.ok()) return 1;
    vk_index_get_stats(ix.handle(), &s2);
    std::queue<std::unique_ptr<EntriesFetcherBase>> fq3;
    fq3.push(fetcher_of(n, 1, false));
    auto f3 = ix.BuildFilter(fq3, [](const InternedStringPtr &key) { return (key->Str().back() - '0') % 2 == 0; });   // IsUnsolvedQuery: evaluated per fetched key
    if (!f3.ok()) return 1;
    for (size_t i = 0; i < 3; ++i) {
      absl::string_view qs(reinterpret_cast<const char *>(q.data() + i * dim), dim * 4);
      auto r1 = ix.Search(qs, k, tok, f1.value(), 64);
      auto r3 = ix.Search(qs, k, tok, f3.value(), 64);
      if (!r1.ok() || !r3.ok()) return 1;
      print_result((std::string(name) + " fetch q" + std::to_string(i)).c_str(), r1.value());
      print_result((std::string(name) + " fetchpred q" + std::to_string(i)).c_str(), r3.value());
      Latch l;
      if (!ix.SearchAsync(qs, k, tok, f2.value(), 64, false, [&l](absl::StatusOr<std::vector<Neighbor>> r) { l.result = std::move(r); l.done.store(1, std::memory_order_release); }).ok()) return 1;
      l.wait();
      if (!l.result.ok()) return 1;
      print_result((std::string(name) + " fetchasync q" + std::to_string(i)).c_str(), l.result.value());
    }
    if (!ix.OnWritePhaseEnd().ok()) return 1;                              // a write phase: cached filters are stale
    std::queue<std::unique_ptr<EntriesFetcherBase>> fq4;
    fq4.push(fetcher_of(n, 2, false));
    auto f4 = ix.BuildFilter(fq4, nullptr, "@parity:{even}");
    if (!f4.ok()) return 1;
    vk_index_get_stats(ix.handle(), &s3);
    printf("%s filters allowed %llu / %llu; built %llu hits %llu misses %llu | cached: built +%llu hits +%llu | after a write phase: built +%llu misses +%llu\n", name,
           (unsigned long long)f1.value().allowed(), (unsigned long long)f3.value().allowed(), (unsigned long long)(s1.filters_built - s0.filters_built),
           (unsigned long long)(s1.filter_cache_hits - s0.filter_cache_hits), (unsigned long long)(s1.filter_cache_misses - s0.filter_cache_misses),
           (unsigned long long)(s2.filters_built - s1.filters_built), (unsigned long long)(s2.filter_cache_hits - s1.filter_cache_hits),
           (unsigned long long)(s3.filters_built - s2.filters_built) - 1, (unsigned long long)(s3.filter_cache_misses - s2.filter_cache_misses));
  }
  // a cancelled token: CancelledError without partial results, an answer (possibly empty) with them
  cancel::Token dead = std::make_shared<Token>();
  dead->Cancel();
  absl::string_view q0(reinterpret_cast<const char *>(q.data()), dim * 4);
  auto rc = ix.Search(q0, k, dead, nullptr, 64, false);
  auto rp = ix.Search(q0, k, dead, nullptr, 64, true);
  printf("%s cancelled: %s / partial %s\n", name, rc.ok() ? "ok" : (rc.status().code() == absl::StatusCode::kCancelled ? "CancelledError" : "other"),
         rp.ok() ? "ok" : "error");
  // modify = same label again; distance of one record; remove
  absl::string_view rec1(reinterpret_cast<const char *>(x.data() + 1 * dim), dim * 4);
  if (!ix.MockModify(0, rec1, std::make_shared<InternedString>(std::string(rec1))).ok()) return 1;
  auto d = ix.MockDistance(0, rec1);
  uint32_t bits = 0;
  if (d.ok()) memcpy(&bits, &d.value().first, 4);
  printf("%s distance(0 := row 1, row 1) %08x label %zu\n", name, bits, d.ok() ? d.value().second : (size_t)0);
  if (!ix.MockRemove(2).ok()) return 1;
  auto r2 = ix.Search(absl::string_view(reinterpret_cast<const char *>(x.data() + 2 * dim), dim * 4), k, tok, nullptr, 64);
  if (!r2.ok()) return 1;
  print_result((std::string(name) + " after remove(2)").c_str(), r2.value());
  printf("%s GetValue(1) %s, GetValue(2) %s, IsVectorMatch %d\n", name, ix.MockGetValue(1) && memcmp(ix.MockGetValue(1), rec1.data(), dim * 4) == 0 ? "stored row" : "MISMATCH",
         ix.MockGetValue(2) == nullptr ? "null" : "present", (int)ix.MockIsVectorMatch(1, std::make_shared<InternedString>(std::string(rec1))));
  data_model::VectorIndex proto = definition;     // (ToProtoImpl fills the algorithm's own fields)
  ix.MockToProto(&proto);
  g_info.clear();
  const int n_info = ix.MockInfo(nullptr);
  printf("%s info %d fields:", name, n_info);
  for (size_t i = 0; i + 1 < g_info.size() && i < 12; i += 2) printf(" %s=%s", g_info[i].c_str(), g_info[i + 1].c_str());
  for (size_t i = 0; i + 1 < g_info.size(); i += 2)
    if (g_info[i] == "gpu_searches") printf(" gpu_searches=%s", g_info[i + 1].c_str());
  printf("\n");
  if ((size_t)n_info != g_info.size()) return 1;
  // SaveIndex -> LoadIndex: the reloaded index answers like the live one
  Chunks ch;
  if (!ix.MockSave(ch).ok()) return 1;
  vk_index *re = nullptr;
  vk_index_params p = ix.params();
  if (vk_index_load(&p, read_chunk, &ch, &re) != VK_OK) { printf("%s load failed: %s\n", name, vk_last_error()); return 1; }
  vk_index_stats s0, s1;
  vk_index_get_stats(ix.handle(), &s0);
  vk_index_get_stats(re, &s1);
  printf("%s save/load chunks %zu count %llu -> %llu\n", name, ch.c.size(), (unsigned long long)s0.count, (unsigned long long)s1.count);
  vk_index_destroy(re);
  // LoadFromRDB (vector_flat.cc:100-124 / vector_hnsw.cc:135-166; index_schema.cc:175,200): the saved chunks through the
  // mock SupplementalContentChunkIter; the loaded index answers like the live one, knows its largest label and got its
  // vectors back through VectorBase::TrackVector
  {
    absl::string_view q0(reinterpret_cast<const char *>(q.data()), dim * 4);
    auto live = ix.Search(q0, k, tok, VkFilterRef(), 64);
    if (!live.ok()) return 1;
    print_result((std::string(name) + " final q0").c_str(), live.value());
    HashAttributeDataType adt;
    auto loaded = Ix::LoadFromRDB(nullptr, &adt, proto, "v", SupplementalContentChunkIter(ch.c));
    if (!loaded.ok()) { printf("%s LoadFromRDB failed: %s\n", name, loaded.status().message().c_str()); return 1; }
    Ix &lx = *loaded.value();
    auto rl = lx.Search(q0, k, tok, VkFilterRef(), 64);
    if (!rl.ok()) return 1;
    print_result((std::string(name) + " loadrdb q0").c_str(), rl.value());
    const char *v1 = lx.MockGetValue(1);
    printf("%s loadrdb count %zu max_label %llu GetValue(1) %s\n", name, lx.GetLabelCount(), (unsigned long long)lx.GetMaxInternalLabel(),
           v1 && memcmp(v1, x.data() + 1 * dim, dim * 4) == 0 ? "stored row" : "MISMATCH");
  }
  return 0;
}

// a chunk file written by the Python test: [u64 length][bytes] ...
static std::vector<std::string> read_chunks(const char *path) {
  std::vector<std::string> out;
  FILE *f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot read ") + path);
  uint64_t len;
  while (fread(&len, 8, 1, f) == 1) {
    std::string c((size_t)len, '\0');
    if (len && fread(&c[0], 1, (size_t)len, f) != len) throw std::runtime_error("truncated chunk file");
    out.push_back(std::move(c));
  }
  fclose(f);
  return out;
}

// `load flat|hnsw <chunks> <dim> <m> <queries> <nq>`: a stream ASSEMBLED BY HAND in the reference's layout
// (tests/helpers/streams.py) through LoadFromRDB
template <class Ix>
static int load_only(const char *kind, const char *chunk_path, size_t dim, size_t m, const char *q_path, size_t nq) {
  data_model::VectorIndex proto;
  proto.dimension_count_ = (uint32_t)dim;
  proto.initial_cap_ = 16;
  proto.distance_metric_ = data_model::DISTANCE_METRIC_L2;
  proto.hnsw_.m_ = (uint32_t)m;
  proto.hnsw_.ef_construction_ = 20;
  HashAttributeDataType adt;
  auto loaded = Ix::LoadFromRDB(nullptr, &adt, proto, "v", SupplementalContentChunkIter(read_chunks(chunk_path)));
  if (!loaded.ok()) { printf("hand %s LoadFromRDB failed: %s\n", kind, loaded.status().message().c_str()); return 1; }
  Ix &ix = *loaded.value();
  printf("hand %s count %zu max_label %llu capacity %zu\n", kind, ix.GetLabelCount(), (unsigned long long)ix.GetMaxInternalLabel(), ix.GetCapacity());
  const std::vector<float> q = read_f32(q_path, nq * dim);
  cancel::Token tok = std::make_shared<Token>();
  for (size_t i = 0; i < nq; ++i) {
    auto r = ix.Search(absl::string_view(reinterpret_cast<const char *>(q.data() + i * dim), dim * 4), 5, tok, VkFilterRef(), 16);
    if (!r.ok()) { printf("hand %s search failed: %s\n", kind, r.status().message().c_str()); return 1; }
    print_result((std::string("hand ") + kind + " q" + std::to_string(i)).c_str(), r.value());
  }
  printf("adaptor ok\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 8 && std::string(argv[1]) == "load") {
    const bool flat = std::string(argv[2]) == "flat";
    const size_t dim = (size_t)atoi(argv[4]), m = (size_t)atoi(argv[5]), nq = (size_t)atoi(argv[7]);
    return flat ? load_only<VectorGpuFlat<float>>("flat", argv[3], dim, m, argv[6], nq) : load_only<VectorGpuHNSW<float>>("hnsw", argv[3], dim, m, argv[6], nq);
  }
  if (argc < 3) return 2;
  const size_t n = 3000, dim = 24;
  const std::vector<float> x = read_f32(argv[1], n * dim), q = read_f32(argv[2], 3 * dim);
  data_model::VectorIndex proto;
  proto.dimension_count_ = dim;
  proto.initial_cap_ = 1000;
  proto.distance_metric_ = data_model::DISTANCE_METRIC_L2;
  proto.hnsw_.ef_construction_ = 100;
  auto flat = VectorGpuFlat<float>::Create(proto, "v", data_model::ATTRIBUTE_DATA_TYPE_HASH);
  if (!flat.ok()) { printf("create failed: %s\n", flat.status().message().c_str()); return 1; }
  if (run(*flat.value(), "flat", x, q, n, dim, proto)) return 1;
  auto hnsw = VectorGpuHNSW<float>::Create(proto, "v", data_model::ATTRIBUTE_DATA_TYPE_HASH, false, 1024);
  if (!hnsw.ok()) { printf("create failed: %s\n", hnsw.status().message().c_str()); return 1; }
  if (run(*hnsw.value(), "hnsw", x, q, n, dim, proto)) return 1;
  {   // the token watcher: a token that goes up is relayed to its request's cancel word within a few ticks, with or without a
      // known deadline; an unregistered request is left alone
    volatile int w1 = 0, w2 = 0, w3 = 0;
    auto t1 = std::make_shared<Token>(), t2 = std::make_shared<Token>(), t3 = std::make_shared<Token>();
    auto &watch = VkTokenWatch::Instance();
    auto h1 = watch.Register(t1, &w1, std::nullopt);
    auto h2 = watch.Register(t2, &w2, std::chrono::steady_clock::now() + std::chrono::milliseconds(5));
    auto h3 = watch.Register(t3, &w3, std::nullopt);
    watch.Unregister(h3);
    t1->Cancel(); t2->Cancel(); t3->Cancel();
    const auto t0 = std::chrono::steady_clock::now();
    while ((!w1 || !w2) && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(500)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    watch.Unregister(h1);
    watch.Unregister(h2);
    printf("watch raised %d %d untouched %d within %s\n", (int)w1, (int)w2, (int)!w3, us < 100000 ? "100 ms" : "TOO LONG");
  }
  printf("adaptor ok\n");
  return 0;
}
