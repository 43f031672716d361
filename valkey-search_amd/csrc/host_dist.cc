// host_dist.cc -- see host_dist.hpp.  Built with -ffp-contract=off: only fmaf may fuse.
#include "host_dist.hpp"

namespace vk {
namespace {

#define VK_AINLINE static inline __attribute__((always_inline))

// 16 accumulator lanes, lane l fed elements l, l+16, l+32, ... in order; a partial last
// block multiplies zeros (the AVX-512 masked load), which leaves the lanes unchanged.
template <bool kL2>
VK_AINLINE float dist16_body(const float *a, const float *b, size_t n) {
  float acc[16];
  for (int l = 0; l < 16; ++l) acc[l] = 0.0f;
  const size_t full = n / 16;
  for (size_t c = 0; c < full; ++c) {
    const float *pa = a + 16 * c, *pb = b + 16 * c;
    for (int l = 0; l < 16; ++l) {
      if (kL2) {
        float d = pa[l] - pb[l];
        acc[l] = __builtin_fmaf(d, d, acc[l]);
      } else {
        acc[l] = __builtin_fmaf(pa[l], pb[l], acc[l]);
      }
    }
  }
  const size_t rem = n - 16 * full;
  if (rem) {
    float ta[16], tb[16];
    for (int l = 0; l < 16; ++l) ta[l] = tb[l] = 0.0f;
    for (size_t l = 0; l < rem; ++l) { ta[l] = a[16 * full + l]; tb[l] = b[16 * full + l]; }
    for (int l = 0; l < 16; ++l) {
      if (kL2) {
        float d = ta[l] - tb[l];
        acc[l] = __builtin_fmaf(d, d, acc[l]);
      } else {
        acc[l] = __builtin_fmaf(ta[l], tb[l], acc[l]);
      }
    }
  }
  // halving tree (l,l+8) -> (l,l+4) -> (l,l+2) -> (0,1)
  float t8[8], t4[4];
  for (int l = 0; l < 8; ++l) t8[l] = acc[l + 8] + acc[l];
  for (int l = 0; l < 4; ++l) t4[l] = t8[l + 4] + t8[l];
  const float sum = (t4[0] + t4[2]) + (t4[1] + t4[3]);
  if (kL2) return sum;
  return (float)(1.0 - (double)sum);
}

float ip_generic(const float *a, const float *b, size_t n) { return dist16_body<false>(a, b, n); }
float l2_generic(const float *a, const float *b, size_t n) { return dist16_body<true>(a, b, n); }
__attribute__((target("avx2,fma"))) float ip_avx2(const float *a, const float *b, size_t n) { return dist16_body<false>(a, b, n); }
__attribute__((target("avx2,fma"))) float l2_avx2(const float *a, const float *b, size_t n) { return dist16_body<true>(a, b, n); }
__attribute__((target("avx512f,avx512vl,fma"))) float ip_avx512(const float *a, const float *b, size_t n) { return dist16_body<false>(a, b, n); }
__attribute__((target("avx512f,avx512vl,fma"))) float l2_avx512(const float *a, const float *b, size_t n) { return dist16_body<true>(a, b, n); }

int cpu_level() {
  static int level = -1;
  if (level < 0) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) level = 2;
    else if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) level = 1;
    else level = 0;
  }
  return level;
}

}  // namespace

host_dist_fn host_distance_ip() {
  int l = cpu_level();
  return l == 2 ? ip_avx512 : l == 1 ? ip_avx2 : ip_generic;
}
host_dist_fn host_distance_l2() {
  int l = cpu_level();
  return l == 2 ? l2_avx512 : l == 1 ? l2_avx2 : l2_generic;
}
const char *host_distance_path() {
  int l = cpu_level();
  return l == 2 ? "avx512f" : l == 1 ? "avx2+fma" : "generic";
}

}  // namespace vk
