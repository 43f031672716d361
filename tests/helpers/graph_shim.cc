// Test helper: exposes the PRODUCT's host-side HNSW builder (valkey-search_amd/csrc/hnsw_graph.cc,
// pure host C++) through a flat C API so the CPU test-suite can compare the graph it builds with the
// oracle's, without a GPU.  Not part of the product library.
#include "../../valkey-search_amd/csrc/hnsw_graph.hpp"

using vk::HnswGraph;

extern "C" {
void *gs_new(uint32_t dim, int l2, size_t max_elements, size_t M, size_t efc, size_t seed, int replace) {
  return new HnswGraph(dim, l2 != 0, max_elements, M, efc, seed, replace != 0);
}
void gs_free(void *g) { delete static_cast<HnswGraph *>(g); }
int gs_add(void *g, const float *row, uint64_t label) {
  uint32_t id;
  return static_cast<HnswGraph *>(g)->add(row, label, &id).code;
}
// the slot the label landed in (a new slot, its own on an update, or a tombstoned one taken over); -1 on error
long gs_add_at(void *g, const float *row, uint64_t label) {
  uint32_t id;
  return static_cast<HnswGraph *>(g)->add(row, label, &id).code ? -1L : (long)id;
}
int gs_mark_delete(void *g, uint64_t label) { return static_cast<HnswGraph *>(g)->mark_delete(label).code; }
int gs_resize(void *g, size_t n) { return static_cast<HnswGraph *>(g)->resize(n).code; }
size_t gs_count(void *g) { return static_cast<HnswGraph *>(g)->count(); }
int gs_max_level(void *g) { return static_cast<HnswGraph *>(g)->max_level(); }
uint32_t gs_entry_point(void *g) { return static_cast<HnswGraph *>(g)->entry_point(); }
int gs_level_of(void *g, uint32_t id) { return static_cast<HnswGraph *>(g)->level_of(id); }
uint64_t gs_label_of(void *g, uint32_t id) { return static_cast<HnswGraph *>(g)->label_of(id); }
int gs_is_deleted(void *g, uint32_t id) { return static_cast<HnswGraph *>(g)->is_deleted(id); }
size_t gs_links(void *g, uint32_t id, int level, uint32_t *out) {
  HnswGraph *h = static_cast<HnswGraph *>(g);
  const uint32_t *ll = level == 0 ? h->links0(id) : h->upper(id, level);
  size_t n = ll[0] & 0xFFFFu;
  for (size_t i = 0; i < n; ++i) out[i] = ll[1 + i];
  return n;
}
#ifdef VK_PROFILE_LOCKS
uint64_t gs_spin_cycles() { return vk::g_spin_wait_cycles.load(); }
uint64_t gs_spin_waits() { return vk::g_spin_waits.load(); }
#endif
const char *gs_dist_path() { return vk::host_distance_path(); }
float gs_distance(int l2, const float *a, const float *b, size_t n) {
  return (l2 ? vk::host_distance_l2() : vk::host_distance_ip())(a, b, n);
}
}
