// What does a small kernel see right behind a long matrix-core kernel?  Kernel A keeps the MFMA pipes and the HBM busy for a
// few milliseconds (the power state of the candidate filter's final pass); kernel B is a fixed chain of dependent integer
// adds (cycles, not bytes).  B is timed on its own, right behind A, and a second time behind that.
//   hipcc --offload-arch=gfx950 -O3 scripts/clock_after_load.hip -o scripts/clock_after_load
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void heavy(const float4 *rows, size_t n4, float *out, int iters) {
  f32x16 acc[4] = {};
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  float4 s = make_float4(0, 0, 0, 0);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    const float4 x = rows[i % n4];
    i += (size_t)gridDim.x * blockDim.x;
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  float v = s.x + s.y + s.z + s.w;
  for (int t = 0; t < 4; ++t) v += acc[t][0];
  if (v == 123.456f) out[0] = v;
}
__global__ void chain(uint32_t *out, int n) {
  uint32_t x = threadIdx.x;
  for (int i = 0; i < n; ++i) x = x * 1664525u + 1013904223u;   // dependent: one v_mad per step
  if (x == 12345u) out[0] = x;
}
int main() {
  const size_t bytes = (size_t)8 << 30;
  float4 *rows; float *out; uint32_t *o2;
  hipMalloc(&rows, bytes); hipMemset(rows, 0, bytes); hipMalloc(&out, 4); hipMalloc(&o2, 4);
  hipEvent_t e[6]; for (auto &x : e) hipEventCreate(&x);
  const int n = 40000;
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e[0]); hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, o2, n); hipEventRecord(e[1]);
    hipEventSynchronize(e[1]);
    float alone; hipEventElapsedTime(&alone, e[0], e[1]);
    hipEventRecord(e[0]);
    hipLaunchKernelGGL(heavy, dim3(256 * 2), dim3(256), 0, 0, rows, bytes / 16, out, 12000);
    hipEventRecord(e[1]);
    hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, o2, n); hipEventRecord(e[2]);
    hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, o2, n); hipEventRecord(e[3]);
    hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, o2, n); hipEventRecord(e[4]);
    hipEventSynchronize(e[4]);
    float h, b1, b2, b3;
    hipEventElapsedTime(&h, e[0], e[1]); hipEventElapsedTime(&b1, e[1], e[2]); hipEventElapsedTime(&b2, e[2], e[3]); hipEventElapsedTime(&b3, e[3], e[4]);
    printf("chain of %d dependent v_mad: alone %.1f us | heavy kernel %.2f ms, then chain %.1f us, %.1f us, %.1f us\n", n, alone * 1e3, h, b1 * 1e3, b2 * 1e3, b3 * 1e3);
  }
  return 0;
}
