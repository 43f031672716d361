// mfma_peak.hip -- what v_mfma_f32_32x32x2_f32 sustains on gfx950 with K4's register shape
// (16 independent f32x16 accumulators per wave, one wave per SIMD), with and without the
// per-tile epilogue.  Prints TFLOP/s against the 157.3 TFLOP/s dense f32 matrix peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int kEpilogue>
__global__ __launch_bounds__(256, 1) void mfma_loop(float *out, int tiles, int stages, float a0, float b0) {
  f32x16 acc[16];
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  float keep = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 16; ++l) acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, zero, 0, 0, 0);
    for (int s = 1; s < stages; ++s) {
#pragma unroll
      for (int l = 0; l < 16; ++l) acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[l], 0, 0, 0);
    }
    if (kEpilogue) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) v[l] = acc[l][r];
#pragma unroll
        for (int l = 0; l < 8; ++l) v[l] = v[l + 8] + v[l];
#pragma unroll
        for (int l = 0; l < 4; ++l) v[l] = v[l + 4] + v[l];
        const float d = 1.0f - ((v[0] + v[2]) + (v[1] + v[3]));
        if (d < keep) keep = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int l = 0; l < 16; ++l) keep += acc[l][0];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

int main() {
  float *out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int stages = 24;
  for (int blocks : {256, 512}) {
    for (int ep = 0; ep < 2; ++ep) {
      const int tiles = 2400 * 256 / blocks;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (ep) mfma_loop<1><<<blocks, 256>>>(out, tiles, stages, 1.f, 2.f);
        else mfma_loop<0><<<blocks, 256>>>(out, tiles, stages, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * tiles * stages * 16 * 32 * 32 * 2 * 2;
        if (rep) printf("blocks=%d epilogue=%d: %.3f ms, %.1f TFLOP/s (%.1f%% of 157.3)\n", blocks, ep, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
      }
    }
  }
  return 0;
}
