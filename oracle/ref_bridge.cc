// ref_bridge.cc -- ORACLE build helper (test infrastructure only).
// Compiled ONLY where /root/reference exists (oracle/Makefile `ref`).  Contains no
// algorithm: it includes the reference's hnswlib<->SimSIMD bridge header
// (third_party/hnswlib/simsimd.h, the only hnswlib header that needs nothing
// outside the tree) and exports its two inline distance functions -- the exact
// `fstdistfunc_` the FLAT and HNSW backends call -- plus the dispatched capability
// mask, so the oracle's vko_distance() can be pinned against them.
#include "third_party/hnswlib/simsimd.h"

extern "C" {
__attribute__((visibility("default"))) float ref_InnerProductDistanceSimsimd(const float *a, const float *b, size_t dim) {
  return InnerProductDistanceSimsimd(a, b, &dim);
}
__attribute__((visibility("default"))) float ref_L2SqrSimsimd(const float *a, const float *b, size_t dim) {
  return L2SqrSimsimd(a, b, &dim);
}
__attribute__((visibility("default"))) unsigned ref_capabilities(void) { return (unsigned)simsimd_capabilities(); }
}
