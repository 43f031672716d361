/* ref_shim.c -- ORACLE build helper (test infrastructure only).
 * Compiled ONLY where /root/reference exists (oracle/Makefile `ref`).  Contains
 * no algorithm: it includes the reference's SimSIMD header and re-exports the
 * per-ISA f32 kernels, which are `static inline` there (types.h:35), so tests can
 * compare each restated summation order with the real kernel bit for bit. */
#define SIMSIMD_NATIVE_F16 0
#define SIMSIMD_NATIVE_BF16 0
#define SIMSIMD_TARGET_NEON 0
#define SIMSIMD_TARGET_SVE 0
#define SIMSIMD_TARGET_HASWELL 1
#define SIMSIMD_TARGET_SKYLAKE 1
#define SIMSIMD_TARGET_ICE 0
#define SIMSIMD_TARGET_GENOA 0
#define SIMSIMD_TARGET_SAPPHIRE 0
#include "third_party/simsimd/include/simsimd/simsimd.h"

#define EXPORT __attribute__((visibility("default")))
EXPORT void ref_dot_f32_serial(const float *a, const float *b, size_t n, double *r) { simsimd_dot_f32_serial(a, b, n, r); }
EXPORT void ref_dot_f32_haswell(const float *a, const float *b, size_t n, double *r) { simsimd_dot_f32_haswell(a, b, n, r); }
EXPORT void ref_dot_f32_skylake(const float *a, const float *b, size_t n, double *r) { simsimd_dot_f32_skylake(a, b, n, r); }
EXPORT void ref_l2sq_f32_serial(const float *a, const float *b, size_t n, double *r) { simsimd_l2sq_f32_serial(a, b, n, r); }
EXPORT void ref_l2sq_f32_haswell(const float *a, const float *b, size_t n, double *r) { simsimd_l2sq_f32_haswell(a, b, n, r); }
EXPORT void ref_l2sq_f32_skylake(const float *a, const float *b, size_t n, double *r) { simsimd_l2sq_f32_skylake(a, b, n, r); }
