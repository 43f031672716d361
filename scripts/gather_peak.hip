// Ceiling of the HNSW access pattern without the graph: random 3 KB rows from a large table, a quad of lanes per
// row (16 rows per wave at a time), kBatch 16-B pieces in flight per lane -- the loop of quad_row_distance with no
// dependency between rows.   hipcc --offload-arch=gfx950 -O3 scripts/gather_peak.hip -o scripts/gather_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int kBatch>
__device__ __forceinline__ float body(const float4 *rows, const uint32_t *idx, uint32_t n, uint32_t f4_per_row) {
  const int lane = threadIdx.x & 63, j = lane & 3, rq = lane >> 2;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  for (uint32_t r = wave * 16; r < n; r += waves * 16) {
    const float4 *base = rows + (size_t)idx[r + rq] * f4_per_row;
    for (uint32_t c = 0; c < f4_per_row / 4; c += kBatch) {
      float4 x[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) x[u] = base[(c + u) * 4 + j];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
  }
  return acc.x + acc.y + acc.z + acc.w;
}
// the same gather with the HNSW kernel's visited-set traffic next to it: per 16 rows, 23 lanes do one atomicOr (with
// return) on a random word of a bitmap region of `words` words (32 link checks per ~22 evaluated rows)
__global__ __launch_bounds__(256, 4) void g8_atomic(const float4 *rows, const uint32_t *idx, uint32_t n, uint32_t f4_per_row,
                                                    uint32_t *bitmap, uint64_t words, float *out) {
  const int lane = threadIdx.x & 63, j = lane & 3, rq = lane >> 2;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  uint32_t seen = 0;
  for (uint32_t r = wave * 16; r < n; r += waves * 16) {
    if (lane < 23) {
      const uint64_t h = ((uint64_t)idx[(r + lane * 7919u) % n] * 2654435761ull + lane) % words;
      seen += atomicOr(&bitmap[h], 1u << (lane & 31)) & 1u;
    }
    const float4 *base = rows + (size_t)(idx[r + rq] + (seen & 0u)) * f4_per_row;   // (the gather waits for the atomics, like the kernel)
    for (uint32_t c = 0; c < f4_per_row / 4; c += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = base[(c + u) * 4 + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
  }
  const float v = acc.x + acc.y + acc.z + acc.w + (float)seen;
  if (v == 123.456f) out[0] = v;
}
__global__ __launch_bounds__(256, 4) void g8(const float4 *rows, const uint32_t *idx, uint32_t n, uint32_t f, float *out) {
  const float v = body<8>(rows, idx, n, f);
  if (v == 123.456f) out[0] = v;
}
__global__ __launch_bounds__(256, 2) void g24(const float4 *rows, const uint32_t *idx, uint32_t n, uint32_t f, float *out) {
  const float v = body<24>(rows, idx, n, f);
  if (v == 123.456f) out[0] = v;
}

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? atoll(argv[1]) : 10000000, D = 768;
  const uint32_t M = 1u << 22;
  float4 *rows; uint32_t *idx; float *out;
  if (hipMalloc(&rows, N * D * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(rows, 0, N * D * 4);
  hipMalloc(&idx, M * 4); hipMalloc(&out, 4);
  std::vector<uint32_t> h(M);
  uint64_t s = 88172645463325252ull;
  for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % N); }
  hipMemcpy(idx, h.data(), M * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int variant = 0; variant < 2; ++variant)
    for (int blocks_per_cu : {1, 2, 4}) {
      if (variant == 1 && blocks_per_cu > 2) continue;
      const int blocks = 256 * blocks_per_cu;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (variant == 0) hipLaunchKernelGGL(g8, dim3(blocks), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), out);
        else hipLaunchKernelGGL(g24, dim3(blocks), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
      }
      printf("rows=%zu kBatch=%d waves/CU=%d: %.2f ms = %.0f GB/s\n", N, variant ? 24 : 8, blocks_per_cu * 4, ms,
             (double)M * D * 4 / ms / 1e6);
    }
  for (uint64_t mb : {5000ull, 1000ull, 100ull, 8ull}) {
    uint32_t *bm; const uint64_t words = mb * 1000000ull / 4;
    if (hipMalloc(&bm, words * 4) != hipSuccess) { printf("bitmap alloc failed\n"); return 1; }
    hipMemset(bm, 0, words * 4);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(g8_atomic, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, words, out);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    printf("rows=%zu kBatch=8 waves/CU=16 + 23 atomics per 16 rows on a %llu MB bitmap: %.2f ms = %.0f GB/s of rows\n", N,
           (unsigned long long)mb, ms, (double)M * D * 4 / ms / 1e6);
    hipFree(bm);
  }
  return 0;
}
