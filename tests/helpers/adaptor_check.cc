// Compile-and-run check of include/vk_vector_adaptor.h -- the VectorBase-derived binding -- against the MOCK of the
// module's interface (tests/helpers/mock_valkey_search.h).  `-Wall -Werror` on every CPU run (tests/test_abi_symbols.py:
// the overrides must match the virtuals' signatures); on a GPU (tests/test_facade_gpu.py) it is run: AddRecord through the
// resize-and-retry loop, Search with ef / filter functor / cancellation token / partial results, Modify / Remove,
// ComputeDistanceFromRecord, GetValue, ToProto, RespondWithInfo, SaveIndex -> vk_index_load.  One line per check; the
// Python test compares the search lines with the ctypes binding's answers on the same data.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
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
static int run(Ix &ix, const char *name, const std::vector<float> &x, const std::vector<float> &q, size_t n, size_t dim) {
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
  data_model::VectorIndex proto;
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
  return 0;
}

int main(int argc, char **argv) {
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
  if (run(*flat.value(), "flat", x, q, n, dim)) return 1;
  auto hnsw = VectorGpuHNSW<float>::Create(proto, "v", data_model::ATTRIBUTE_DATA_TYPE_HASH, false, 1024);
  if (!hnsw.ok()) { printf("create failed: %s\n", hnsw.status().message().c_str()); return 1; }
  if (run(*hnsw.value(), "hnsw", x, q, n, dim)) return 1;
  printf("adaptor ok\n");
  return 0;
}
