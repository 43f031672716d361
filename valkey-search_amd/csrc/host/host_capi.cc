// host_capi.cc -- flat C shim over the host mirror classes so the Python test-suite can replay the
// reference's unit-test scenarios (testing/vector_test.cc, testing/search_test.cc) against them.
// Test harness surface only; the product ABI is include/vk_index.h.
#include <string.h>

#include "vector_index.h"

using namespace vsa;
using namespace vsa::indexes;

namespace {
struct Handle {
  std::shared_ptr<VectorBase> base;
  std::shared_ptr<VectorFlat<float>> flat;
  std::shared_ptr<VectorHNSW<float>> hnsw;
  std::string err;
};
thread_local std::string g_err;
int code_of(const Status &s) {
  g_err = s.message();
  return (int)s.code();
}
}  // namespace

extern "C" {
const char *vsa_last_error() { return g_err.c_str(); }

void *vsa_flat_create(int dim, int metric, uint64_t initial_cap, uint32_t block_size) {
  auto r = VectorFlat<float>::Create(FlatParams{dim, (DistanceMetric)metric, initial_cap, block_size});
  if (!r.ok()) { g_err = r.status().message(); return nullptr; }
  auto *h = new Handle;
  h->flat = r.value();
  h->base = h->flat;
  return h;
}
void *vsa_hnsw_create(int dim, int metric, uint64_t initial_cap, uint32_t m, uint32_t efc, uint32_t ef, uint32_t block) {
  auto r = VectorHNSW<float>::Create(HnswParams{dim, (DistanceMetric)metric, initial_cap, m, efc, ef, block, false});
  if (!r.ok()) { g_err = r.status().message(); return nullptr; }
  auto *h = new Handle;
  h->hnsw = r.value();
  h->base = h->hnsw;
  return h;
}
void vsa_destroy(void *p) { delete static_cast<Handle *>(p); }

// returns status code; *result = RecordResult when ok
int vsa_add_record(void *p, const char *key, const void *rec, uint64_t len, int *result) {
  auto r = static_cast<Handle *>(p)->base->AddRecord(key, std::string_view((const char *)rec, len));
  if (!r.ok()) return code_of(r.status());
  *result = (int)r.value();
  return 0;
}
int vsa_modify_record(void *p, const char *key, const void *rec, uint64_t len, int *result) {
  auto r = static_cast<Handle *>(p)->base->ModifyRecord(key, std::string_view((const char *)rec, len));
  if (!r.ok()) return code_of(r.status());
  *result = (int)r.value();
  return 0;
}
int vsa_remove_record(void *p, const char *key, int *removed) {
  auto r = static_cast<Handle *>(p)->base->RemoveRecord(key);
  if (!r.ok()) return code_of(r.status());
  *removed = r.value() ? 1 : 0;
  return 0;
}
int vsa_is_tracked(void *p, const char *key) { return static_cast<Handle *>(p)->base->IsTracked(key) ? 1 : 0; }
uint64_t vsa_capacity(void *p) { return static_cast<Handle *>(p)->base->GetCapacity(); }
uint64_t vsa_tracked(void *p) { return static_cast<Handle *>(p)->base->GetTrackedKeyCount(); }
int vsa_normalize(void *p) { return static_cast<Handle *>(p)->base->GetNormalize() ? 1 : 0; }
int vsa_use_prefiltering(void *p, uint64_t est) { return query::UsePreFiltering(est, static_cast<Handle *>(p)->base.get()) ? 1 : 0; }

static int emit(const StatusOr<std::vector<Neighbor>> &r, char *keys, uint64_t keys_cap, float *dist, uint64_t *n) {
  if (!r.ok()) return code_of(r.status());
  uint64_t off = 0, i = 0;
  for (const auto &nb : r.value()) {
    if (off + nb.external_id.size() + 1 > keys_cap) break;
    memcpy(keys + off, nb.external_id.c_str(), nb.external_id.size() + 1);
    off += nb.external_id.size() + 1;
    dist[i++] = nb.distance;
  }
  *n = i;
  return 0;
}

// allowed_keys: NUL-separated list of keys that pass the filter (n_allowed < 0 = no filter)
int vsa_search(void *p, const void *q, uint64_t qlen, uint64_t k, int64_t ef, const char *allowed_keys, int64_t n_allowed,
               int cancelled, int partial_ok, char *keys, uint64_t keys_cap, float *dist, uint64_t *n) {
  Handle *h = static_cast<Handle *>(p);
  std::unordered_map<std::string, bool> allow;
  KeyPredicate pred;
  if (n_allowed >= 0) {
    const char *c = allowed_keys;
    for (int64_t i = 0; i < n_allowed; ++i) { allow[c] = true; c += strlen(c) + 1; }
    pred = [&allow](const std::string &key) { return allow.count(key) != 0; };
  }
  cancel::Token tok = cancel::Make();
  if (cancelled) tok->Cancel();
  std::string_view qv((const char *)q, qlen);
  if (h->flat) return emit(h->flat->Search(qv, k, tok, n_allowed >= 0 ? &pred : nullptr), keys, keys_cap, dist, n);
  return emit(h->hnsw->Search(qv, k, tok, n_allowed >= 0 ? &pred : nullptr,
                              ef > 0 ? std::optional<size_t>((size_t)ef) : std::nullopt, partial_ok != 0),
              keys, keys_cap, dist, n);
}

int vsa_search_prefiltered(void *p, const void *q, uint64_t qlen, uint64_t k, const char *key_list, int64_t n_keys,
                           char *keys, uint64_t keys_cap, float *dist, uint64_t *n) {
  std::vector<std::string> ks;
  const char *c = key_list;
  for (int64_t i = 0; i < n_keys; ++i) { ks.emplace_back(c); c += strlen(c) + 1; }
  return emit(static_cast<Handle *>(p)->base->SearchPrefiltered(std::string_view((const char *)q, qlen), k, ks), keys, keys_cap, dist, n);
}

int vsa_get_value(void *p, const char *key, void *out, uint64_t cap) {
  auto r = static_cast<Handle *>(p)->base->GetValue(key);
  if (!r.ok()) return code_of(r.status());
  memcpy(out, r.value().data(), std::min<uint64_t>(cap, r.value().size()));
  return 0;
}
}
