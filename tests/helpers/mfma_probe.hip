// mfma_probe.hip -- TEST INFRASTRUCTURE (tests/test_mfma_error_gpu.py), not part of libvkindex.so.
//
// The candidate filter's margin (valkey-search_amd/csrc/flat_filter.hip, header: "Error bound") ALLOWS the matrix cores
// D * 2^-22 * sum |x_i q_i| of accumulation error for a chain of v_mfma_f32_32x32x16_{f16,bf16} over D elements.  That
// allowance was an assumption about the instruction.  This probe runs exactly the chain the filter's consumers run --
// first K-step into the constant 0, then K/16 - 1 accumulating steps, same operand layout (lane l holds elements
// (l / 32) * 8 .. +8 of row / column l % 32 of a K-step) -- on operands the test chooses, and hands the 32 x 32 f32
// results back, so the test can compare them with the exact (f64, from the f16 / bf16 values themselves) dot products.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A [n][32][K] (rows), B [n][32][K] (query columns), C [n][32][32]; 16-bit patterns; one wave per problem
template <bool kBf16>
__global__ __launch_bounds__(64) void mfma_chain_kernel(const uint16_t *A, const uint16_t *B, uint32_t K, float *C) {
  const uint32_t lane = threadIdx.x, i = lane & 31, g = lane >> 5;
  const uint16_t *a = A + ((size_t)blockIdx.x * 32 + i) * K + g * 8;
  const uint16_t *b = B + ((size_t)blockIdx.x * 32 + i) * K + g * 8;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (uint32_t ks = 0; ks < K / 16; ++ks) {
    const f16x8 x = *reinterpret_cast<const f16x8 *>(a + ks * 16);
    const f16x8 y = *reinterpret_cast<const f16x8 *>(b + ks * 16);
    if constexpr (kBf16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
  }
  float *c = C + (size_t)blockIdx.x * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[r];   // (row, column) as filter_gate reads them
}

extern "C" int mfma_probe_chain(const uint16_t *hA, const uint16_t *hB, uint32_t n, uint32_t K, int bf16, float *hC) {
  if (K == 0 || K % 16 != 0 || n == 0) return 1;
  const size_t ab = (size_t)n * 32 * K * 2, cb = (size_t)n * 1024 * 4;
  uint16_t *dA = nullptr, *dB = nullptr;
  float *dC = nullptr;
  int rc = 2;
  if (hipMalloc(&dA, ab) == hipSuccess && hipMalloc(&dB, ab) == hipSuccess && hipMalloc(&dC, cb) == hipSuccess &&
      hipMemcpy(dA, hA, ab, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dB, hB, ab, hipMemcpyHostToDevice) == hipSuccess) {
    if (bf16) hipLaunchKernelGGL(mfma_chain_kernel<true>, dim3(n), dim3(64), 0, nullptr, dA, dB, K, dC);
    else hipLaunchKernelGGL(mfma_chain_kernel<false>, dim3(n), dim3(64), 0, nullptr, dA, dB, K, dC);
    if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(hC, dC, cb, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  return rc;
}
