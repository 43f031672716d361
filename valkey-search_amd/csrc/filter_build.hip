// filter_build.hip -- allow-sets built ON THE DEVICE from what the query layer already holds.
//
// valkey-search filters an HNSW search with InlineVectorFilter (src/query/search.cc:103-134): a functor called per visited
// candidate, label -> key -> predicate.  A functor cannot be called from a kernel, so a search carries a bitmap over the
// labels instead.  The module does not have that bitmap -- it has the EntriesFetchers of the predicate
// (search.cc:301-399; tag.cc:383-455 yields the keys of the matched tags), i.e. LISTS of keys / internal ids.  These
// kernels turn such lists (ids in any order, duplicates allowed: an OR of fetchers, search.cc:208-220) or sorted id RUNS
// [first, last] into the bitmap: 8 B per listed id over PCIe and one scatter pass instead of a host sweep over every
// label of the index.  HBM-bound integer work: one u64 load per id, one atomic OR per id (ids of one fetcher are random
// in label space, so neighbouring lanes seldom share a word); runs are written a word per lane.
#include <algorithm>

#include "kernels.hpp"

namespace vk {
namespace {
constexpr int kThreads = 256;

// (both kernels also COUNT the bits they turn on -- the atomic OR returns the word as it was, so a bit is counted by exactly
//  one of the lanes that set it, duplicates and overlaps included -- as one partial sum per block; the host that waits for the
//  build adds them up: no counter to clear, no popcount pass over the bitmap)
__device__ __forceinline__ void block_partial(unsigned long long c, unsigned long long *partial) {
  __shared__ unsigned long long s_c[kThreads / 64];
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63u) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_c[w];
    partial[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kThreads) void filter_set_ids_kernel(unsigned long long *bits, uint64_t nbits, const uint64_t *ids, uint64_t n,
                                                                   unsigned long long *partial) {
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t id = ids[i];
    if (id < nbits) {   // (labels beyond the bitmap are rejected, vk_index.h allow_nbits)
      const unsigned long long m = 1ull << (id & 63);
      c += (atomicOr(&bits[id >> 6], m) & m) == 0 ? 1u : 0u;
    }
  }
  block_partial(c, partial);
}

// one wave per run: the run's words are dealt to the lanes, the two edge words are masked.  Runs may overlap or share a
// word with their neighbours, hence the atomic.
__global__ __launch_bounds__(kThreads) void filter_set_runs_kernel(unsigned long long *bits, uint64_t nbits, const uint64_t *runs, uint64_t n_runs,
                                                                    unsigned long long *partial) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, waves = ((uint64_t)gridDim.x * kThreads) >> 6;
  unsigned long long c = 0;
  for (uint64_t r = wave; r < n_runs; r += waves) {
    uint64_t lo = runs[2 * r], hi = runs[2 * r + 1];
    if (lo > hi || lo >= nbits) continue;
    if (hi >= nbits) hi = nbits - 1;
    const uint64_t w0 = lo >> 6, w1 = hi >> 6;
    for (uint64_t w = w0 + lane; w <= w1; w += 64) {
      unsigned long long m = ~0ull;
      if (w == w0) m &= ~0ull << (lo & 63);
      if (w == w1) m &= ~0ull >> (63 - (hi & 63));
      c += (unsigned long long)__popcll(m & ~atomicOr(&bits[w], m));
    }
  }
  block_partial(c, partial);
}

__global__ __launch_bounds__(kThreads) void filter_popcount_kernel(const unsigned long long *bits, uint64_t words, unsigned long long *out) {
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < words; i += (uint64_t)gridDim.x * kThreads) c += (unsigned long long)__popcll(bits[i]);
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(out, c);
}

// dst = a OP b over `words` words (0 = and, 1 = or, 2 = and-not): composed predicates whose parts are cached bitmaps.  The
// same pass counts the result's bits -- a partial sum per block, added up by the host that waits for the kernel anyway (a
// combine per FT.SEARCH: a counter to clear, a popcount launch and its copy were two thirds of its 40 us) -- and clears the
// slack word behind the bitmap (the block may be a recycled one).
__global__ __launch_bounds__(kThreads) void filter_combine_kernel(unsigned long long *dst, const unsigned long long *a, const unsigned long long *b,
                                                                   uint64_t words, uint32_t op, unsigned long long *partial) {
  __shared__ unsigned long long s_c[kThreads / 64];
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < words; i += (uint64_t)gridDim.x * kThreads) {
    const unsigned long long x = a[i], y = b[i];
    const unsigned long long r = op == 0 ? (x & y) : op == 1 ? (x | y) : (x & ~y);
    dst[i] = r;
    c += (unsigned long long)__popcll(r);
  }
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63u) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_c[w];
    partial[blockIdx.x] = t;
    if (blockIdx.x == 0) dst[words] = 0;
  }
}

// ... and n of them in ONE launch (a batch of FT.SEARCH requests each with its own composed predicate: 1024 launches of 18 us
// were 20 ms of a 190 ms step): blockIdx.y names the item, items[y] = {dst, a, b, op}; counts[y] (zeroed by the caller) gets the
// result's bits, one atomic add per block
struct CombineItem { unsigned long long *dst; const unsigned long long *a, *b; unsigned long long op; };
__global__ __launch_bounds__(kThreads) void filter_combine_batch_kernel(const CombineItem *items, uint64_t words, unsigned long long *counts) {
  __shared__ unsigned long long s_c[kThreads / 64];
  const CombineItem it = items[blockIdx.y];
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < words; i += (uint64_t)gridDim.x * kThreads) {
    const unsigned long long x = it.a[i], y = it.b[i];
    const unsigned long long r = it.op == 0 ? (x & y) : it.op == 1 ? (x | y) : (x & ~y);
    it.dst[i] = r;
    c += (unsigned long long)__popcll(r);
  }
  for (int off = 32; off; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63u) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_c[w];
    if (t) atomicAdd(&counts[blockIdx.y], t);
    if (blockIdx.x == 0) it.dst[words] = 0;
  }
}

inline uint32_t grid_for(uint64_t items) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (items + kThreads - 1) / kThreads), 256 * 8); }
}  // namespace

uint32_t filter_set_ids_blocks(uint64_t n) { return grid_for(n); }
uint32_t filter_set_runs_blocks(uint64_t n_runs) { return grid_for(n_runs * 64); }
hipError_t launch_filter_set_ids(uint64_t *bits, uint64_t nbits, const uint64_t *d_ids, uint64_t n, unsigned long long *d_partial, hipStream_t s) {
  hipLaunchKernelGGL(filter_set_ids_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, reinterpret_cast<unsigned long long *>(bits), nbits, d_ids, n,
                     d_partial);
  return hipGetLastError();
}
hipError_t launch_filter_set_runs(uint64_t *bits, uint64_t nbits, const uint64_t *d_runs, uint64_t n_runs, unsigned long long *d_partial,
                                  hipStream_t s) {
  hipLaunchKernelGGL(filter_set_runs_kernel, dim3(grid_for(n_runs * 64)), dim3(kThreads), 0, s, reinterpret_cast<unsigned long long *>(bits), nbits,
                     d_runs, n_runs, d_partial);
  return hipGetLastError();
}
hipError_t launch_filter_popcount(const uint64_t *bits, uint64_t words, unsigned long long *d_out, hipStream_t s) {
  if (words == 0) return hipSuccess;
  hipLaunchKernelGGL(filter_popcount_kernel, dim3(grid_for(words)), dim3(kThreads), 0, s, reinterpret_cast<const unsigned long long *>(bits), words, d_out);
  return hipGetLastError();
}
uint32_t filter_combine_blocks(uint64_t words) { return grid_for(words); }
hipError_t launch_filter_combine(uint64_t *dst, const uint64_t *a, const uint64_t *b, uint64_t words, uint32_t op, unsigned long long *d_partial,
                                 hipStream_t s) {
  hipLaunchKernelGGL(filter_combine_kernel, dim3(grid_for(words)), dim3(kThreads), 0, s, reinterpret_cast<unsigned long long *>(dst),
                     reinterpret_cast<const unsigned long long *>(a), reinterpret_cast<const unsigned long long *>(b), words, op, d_partial);
  return hipGetLastError();
}

// d_items: [n][4] words = {dst, a, b, op} (device pointers); d_counts: [n], zeroed
hipError_t launch_filter_combine_batch(const uint64_t *d_items, uint32_t n, uint64_t words, unsigned long long *d_counts, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (n > 65535) return hipErrorInvalidValue;
  const uint32_t gx = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (words + kThreads * 4 - 1) / (kThreads * 4)), 64);
  hipLaunchKernelGGL(filter_combine_batch_kernel, dim3(gx, n), dim3(kThreads), 0, s, reinterpret_cast<const CombineItem *>(d_items), words, d_counts);
  return hipGetLastError();
}

}  // namespace vk
