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

// r04: what the visited set could cost instead.  Every wave owns a private 64 KB region of `bitmap` (like the kernel's
// hash tables).  kMode 1: the same 23 returning atomics, agent scope, on the private region; 2: wavefront scope (the
// set is private to the wave: the atomic may be served by this XCD's L2 instead of the memory side); 3: 23 dependent
// 16-B loads that bypass L1 + 16 fire-and-forget stores (LDS-resident bucket counts, ids in HBM buckets); 4: the stores
// alone, nothing waited for (look-ups overlapped with the first round of gathers)
template <int kMode, bool kNt = false>
__global__ __launch_bounds__(256, 4) void g8_private(const float4 *rows, const uint32_t *idx, uint32_t n, uint32_t f4_per_row,
                                                     uint32_t *bitmap, float *out) {
  const int lane = threadIdx.x & 63, j = lane & 3, rq = lane >> 2;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  uint32_t *mine = bitmap + (size_t)wave * 16384;
  float4 acc = make_float4(0, 0, 0, 0);
  uint32_t seen = 0;
  for (uint32_t r = wave * 16; r < n; r += waves * 16) {
    if (lane < 23) {
      const uint32_t h = (uint32_t)(((uint64_t)idx[(r + lane * 7919u) % n] * 2654435761ull + lane) % 16384u);
      if constexpr (kMode == 1) seen += atomicOr(&mine[h], 1u << (lane & 31)) & 1u;
      if constexpr (kMode == 2) seen += __hip_atomic_fetch_or(&mine[h], 1u << (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & 1u;
      if constexpr (kMode == 3) {
        const uint32_t *b = mine + (h & ~3u);
        seen += (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^
                 __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ __hip_atomic_load(b + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u;
      }
      if constexpr (kMode == 3 || kMode == 4)
        if (lane < 16) __hip_atomic_store(&mine[(h * 7u + 5u) & 16383u], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (kMode == 5 || kMode == 6)   // whole 32-B / 64-B aligned pieces instead of 4-B words
        if (lane < 16) {
          uint4 *d = reinterpret_cast<uint4 *>(mine + (((h * 7u + 5u) & 16383u) & ~(kMode == 5 ? 7u : 15u)));
          const uint4 v = make_uint4(r, r + 1, r + 2, r + 3);
#pragma unroll
          for (int u = 0; u < (kMode == 5 ? 2 : 4); ++u) d[u] = v;
        }
    }
    const float4 *base = rows + (size_t)(idx[r + rq] + (seen & 0u)) * f4_per_row;
    for (uint32_t c = 0; c < f4_per_row / 4; c += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (kNt) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(&base[(c + u) * 4 + j]));
          x[u] = make_float4(t[0], t[1], t[2], t[3]);
        }
        else x[u] = base[(c + u) * 4 + j];
      }
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
  {
    uint32_t *bm; const uint64_t words = 4096ull * 16384ull;
    if (hipMalloc(&bm, words * 4) != hipSuccess) { printf("table alloc failed\n"); return 1; }
    hipMemset(bm, 0, words * 4);
    const char *what[] = {"", "23 returning atomics, agent scope", "23 returning atomics, wavefront scope",
                          "23 dependent 16-B loads + 16 stores", "16 stores, nothing waited for", "16 stores of 32 B, nothing waited for",
                          "16 stores of 64 B, nothing waited for", "rows non-temporal, nothing else", "rows non-temporal + 23 returning atomics, agent scope",
                          "rows non-temporal + 16 stores, nothing waited for", "rows non-temporal + 23 dependent 16-B loads + 16 stores"};
    for (int mode = 1; mode <= 10; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (mode == 1) hipLaunchKernelGGL(g8_private<1>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 2) hipLaunchKernelGGL(g8_private<2>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 3) hipLaunchKernelGGL(g8_private<3>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 4) hipLaunchKernelGGL(g8_private<4>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 5) hipLaunchKernelGGL(g8_private<5>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 6) hipLaunchKernelGGL(g8_private<6>, dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 7) hipLaunchKernelGGL((g8_private<0, true>), dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 8) hipLaunchKernelGGL((g8_private<1, true>), dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 9) hipLaunchKernelGGL((g8_private<4, true>), dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        if (mode == 10) hipLaunchKernelGGL((g8_private<3, true>), dim3(1024), dim3(256), 0, 0, rows, idx, M, (uint32_t)(D / 4), bm, out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
      }
      printf("rows=%zu kBatch=8 waves/CU=16, per 16 rows %s on a wave-private 64 KB table: %.2f ms = %.0f GB/s of rows\n", N,
             what[mode], ms, (double)M * D * 4 / ms / 1e6);
    }
    hipFree(bm);
  }
  return 0;
}
