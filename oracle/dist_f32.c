/*
 * dist_f32.c -- ORACLE (test infrastructure only, see vko.h).
 *
 * Lane-exact restatement of SimSIMD 5.0.1's f32 `dot` and `l2sq` kernels in
 * plain C.  "Lane-exact" = the same set of independent f32 accumulators, each
 * fed the same fused-multiply-add chain in the same order, then the same
 * reduction tree, so results are bit-identical to the vector code on any host.
 *
 *   skylake (AVX-512): dot.h:1183-1204, spatial.h:1131-1153 -- 16 accumulators,
 *       masked zero tail, _mm512_reduce_add_ps halving tree
 *       (l,l+8) -> (l,l+4) -> (l,l+2) -> (0,1)   [avx512fintrin.h reduce macro]
 *   haswell (AVX2):    dot.h:839-878,  spatial.h:1069-1088  -- 8 accumulators,
 *       lanes widened to f64 before the horizontal sum, scalar tail added in f64
 *   serial:            dot.h:148-159,  spatial.h:126-137    -- one f32 accumulator
 *
 * The bodies are written once and instantiated under three target attributes;
 * the clone is picked at run time from cpuid so the same .so runs on any x86-64
 * box (the arithmetic is identical in every clone: fmaf is a correctly rounded
 * fused multiply-add whether it is one instruction or the libm routine).
 * Build with -ffp-contract=off: only the explicit fmaf calls may fuse.
 */
#include <math.h>
#include <string.h>

#include "vko.h"

#define VKO_INLINE static inline __attribute__((always_inline))

/* ---- skylake: dot.h:1183-1204 -------------------------------------------- */
VKO_INLINE float vko_reduce16(const float v[16]) {
    /* _mm512_reduce_add_ps: upper 256 + lower 256, upper 128 + lower 128,
     * then {2,3,0,1} shuffle add, then element 0 + element 1 */
    float t8[8], t4[4];
    for (int l = 0; l < 8; ++l) t8[l] = v[l + 8] + v[l];
    for (int l = 0; l < 4; ++l) t4[l] = t8[l + 4] + t8[l];
    float u0 = t4[0] + t4[2];
    float u1 = t4[1] + t4[3];
    return u0 + u1;
}

VKO_INLINE double vko_dot_skylake_body(const float *a, const float *b, size_t n) {
    float acc[16];
    for (int l = 0; l < 16; ++l) acc[l] = 0.0f;
    /* the reference's loop body runs at least once, and once more for a
     * partial tail whose masked-off lanes load +0.0f (fma(0,0,x) == x) */
    size_t full = n / 16;
    for (size_t c = 0; c < full; ++c) {
        const float *pa = a + 16 * c, *pb = b + 16 * c;
        for (int l = 0; l < 16; ++l) acc[l] = __builtin_fmaf(pa[l], pb[l], acc[l]);
    }
    size_t rem = n - 16 * full;
    if (rem || full == 0) {
        float ta[16], tb[16];
        for (int l = 0; l < 16; ++l) ta[l] = tb[l] = 0.0f;
        for (size_t l = 0; l < rem; ++l) { ta[l] = a[16 * full + l]; tb[l] = b[16 * full + l]; }
        for (int l = 0; l < 16; ++l) acc[l] = __builtin_fmaf(ta[l], tb[l], acc[l]);
    }
    return (double)vko_reduce16(acc);
}

/* spatial.h:1131-1153 */
VKO_INLINE double vko_l2sq_skylake_body(const float *a, const float *b, size_t n) {
    float acc[16];
    for (int l = 0; l < 16; ++l) acc[l] = 0.0f;
    size_t full = n / 16;
    for (size_t c = 0; c < full; ++c) {
        const float *pa = a + 16 * c, *pb = b + 16 * c;
        for (int l = 0; l < 16; ++l) {
            float d = pa[l] - pb[l];
            acc[l] = __builtin_fmaf(d, d, acc[l]);
        }
    }
    size_t rem = n - 16 * full;
    if (rem || full == 0) {
        float ta[16], tb[16];
        for (int l = 0; l < 16; ++l) ta[l] = tb[l] = 0.0f;
        for (size_t l = 0; l < rem; ++l) { ta[l] = a[16 * full + l]; tb[l] = b[16 * full + l]; }
        for (int l = 0; l < 16; ++l) {
            float d = ta[l] - tb[l];
            acc[l] = __builtin_fmaf(d, d, acc[l]);
        }
    }
    return (double)vko_reduce16(acc);
}

/* ---- haswell: dot.h:839-878 ------------------------------------------------ */
VKO_INLINE double vko_reduce8_dbl(const float v[8]) {
    /* _mm256_reduce_add_ps_dbl (dot.h:839-862): low/high 128 -> f64, add,
     * then (s0+s2, s1+s3), then hadd */
    double s[4];
    for (int i = 0; i < 4; ++i) s[i] = (double)v[i] + (double)v[i + 4];
    double p0 = s[0] + s[2];
    double p1 = s[1] + s[3];
    return p0 + p1;
}

VKO_INLINE double vko_dot_haswell_body(const float *a, const float *b, size_t n) {
    float acc[8];
    for (int l = 0; l < 8; ++l) acc[l] = 0.0f;
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = __builtin_fmaf(a[i + l], b[i + l], acc[l]);
    double ab = vko_reduce8_dbl(acc);
    for (; i < n; ++i) {
        float p = a[i] * b[i]; /* f32 product, then widened (dot.h:876-877) */
        ab += (double)p;
    }
    return ab;
}

/* spatial.h:1069-1088 */
VKO_INLINE double vko_l2sq_haswell_body(const float *a, const float *b, size_t n) {
    float acc[8];
    for (int l = 0; l < 8; ++l) acc[l] = 0.0f;
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) {
            float d = a[i + l] - b[i + l];
            acc[l] = __builtin_fmaf(d, d, acc[l]);
        }
    double d2 = vko_reduce8_dbl(acc);
    for (; i < n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        d2 += (double)p;
    }
    return d2;
}

/* ---- serial: dot.h:148-159, spatial.h:126-137 (f32 accumulator, in order) --- */
static double vko_dot_serial(const float *a, const float *b, size_t n) {
    float ab = 0.0f;
    for (size_t i = 0; i != n; ++i) {
        float p = a[i] * b[i];
        ab += p;
    }
    return (double)ab;
}
static double vko_l2sq_serial(const float *a, const float *b, size_t n) {
    float d2 = 0.0f;
    for (size_t i = 0; i != n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        d2 += p;
    }
    return (double)d2;
}

/* ---- three clones of each vector-order body -------------------------------- */
#define VKO_CLONES(name)                                                                   \
    static double name##_generic(const float *a, const float *b, size_t n) {               \
        return name##_body(a, b, n);                                                       \
    }                                                                                      \
    __attribute__((target("avx2,fma"))) static double name##_avx2(const float *a,          \
                                                                  const float *b, size_t n) { \
        return name##_body(a, b, n);                                                       \
    }                                                                                      \
    __attribute__((target("avx512f,avx512vl,fma"))) static double name##_avx512(           \
        const float *a, const float *b, size_t n) {                                        \
        return name##_body(a, b, n);                                                       \
    }

VKO_CLONES(vko_dot_skylake)
VKO_CLONES(vko_l2sq_skylake)
VKO_CLONES(vko_dot_haswell)
VKO_CLONES(vko_l2sq_haswell)

typedef double (*vko_kernel_t)(const float *, const float *, size_t);
static vko_kernel_t g_dot[3], g_l2sq[3];
static const char *g_path = 0;

static void vko_pick(void) {
    if (g_path) return;
    __builtin_cpu_init();
    g_dot[VKO_ISA_SERIAL] = vko_dot_serial;
    g_l2sq[VKO_ISA_SERIAL] = vko_l2sq_serial;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) {
        g_dot[VKO_ISA_SKYLAKE] = vko_dot_skylake_avx512;
        g_l2sq[VKO_ISA_SKYLAKE] = vko_l2sq_skylake_avx512;
        g_dot[VKO_ISA_HASWELL] = vko_dot_haswell_avx512;
        g_l2sq[VKO_ISA_HASWELL] = vko_l2sq_haswell_avx512;
        g_path = "avx512f";
    } else if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) {
        g_dot[VKO_ISA_SKYLAKE] = vko_dot_skylake_avx2;
        g_l2sq[VKO_ISA_SKYLAKE] = vko_l2sq_skylake_avx2;
        g_dot[VKO_ISA_HASWELL] = vko_dot_haswell_avx2;
        g_l2sq[VKO_ISA_HASWELL] = vko_l2sq_haswell_avx2;
        g_path = "avx2+fma";
    } else {
        g_dot[VKO_ISA_SKYLAKE] = vko_dot_skylake_generic;
        g_l2sq[VKO_ISA_SKYLAKE] = vko_l2sq_skylake_generic;
        g_dot[VKO_ISA_HASWELL] = vko_dot_haswell_generic;
        g_l2sq[VKO_ISA_HASWELL] = vko_l2sq_haswell_generic;
        g_path = "generic(libm fmaf)";
    }
}

const char *vko_cpu_path(void) {
    vko_pick();
    return g_path;
}

double vko_dot_f32(vko_isa_t isa, const float *a, const float *b, size_t n) {
    vko_pick();
    return g_dot[isa](a, b, n);
}

double vko_l2sq_f32(vko_isa_t isa, const float *a, const float *b, size_t n) {
    vko_pick();
    return g_l2sq[isa](a, b, n);
}

/* hnswlib/simsimd.h:16-34.  `1.0f - distance` with distance a double is
 * evaluated in double and narrowed on return. */
float vko_distance(vko_space_t space, vko_isa_t isa, const float *a, const float *b, size_t n) {
    vko_pick();
    if (space == VKO_SPACE_IP) {
        double distance = g_dot[isa](a, b, n);
        return (float)(1.0 - distance);
    }
    return (float)g_l2sq[isa](a, b, n);
}

/* vector_base.cc:112-124.  Sequential f32; the reference TU is compiled with
 * -ffast-math (valkey_search.cmake:121), so its low bits are build dependent;
 * what the reference's tests pin is the three COSINE score strings, which
 * this form reproduces (tests/test_oracle_kat.py). */
float vko_normalize(float *dst, const float *src, size_t n) {
    float magnitude = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float p = src[i] * src[i];
        magnitude += p;
    }
    magnitude = sqrtf(magnitude);
    float norm = (magnitude == 0.0f) ? 1.0f : (1.0f / magnitude);
    for (size_t i = 0; i < n; i++) dst[i] = norm * src[i];
    return magnitude;
}
