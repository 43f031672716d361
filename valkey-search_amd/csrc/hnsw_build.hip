// hnsw_build.hip -- K9: the level-0 part of HierarchicalNSW::addPoint for a BATCH of new points on
// the device (third_party/hnswlib/hnswalg.h:1523-1650), i.e. the steps that dominate graph
// construction: the efConstruction beam search (searchBaseLayer :255-347 -- served by
// hnsw_search_kernel with k = ef = efConstruction), the neighbour-selection heuristic
// (getNeighborsByHeuristic2 :553-594) and the reverse links with re-pruning of full lists
// (mutuallyConnectNewElement :613-756).
//
// Batch semantics: the points of one batch search the graph as it was before the batch (they do
// not see one another), like concurrent addPoint calls that started at the same moment; a node that
// several new points selected gets ONE re-prune over its old list plus all of them.  The host keeps
// batches small relative to the graph (hnsw_index.cc).  The graph therefore differs from a
// sequential build the same way any multi-threaded hnswlib build does -- recall, not ids, is what
// is pinned (tests/test_hnsw_build_gpu.py).  Upper levels (1/M of the points) stay on the host.
//
// Both kernels: one wave per node, distances 16 rows at a time (one row per quad of lanes) against a
// "query" row staged in LDS, the same lane-exact f32 arithmetic as everywhere else.
#include "device_common.hpp"
#include "kernels.hpp"

namespace vk {

namespace {
constexpr uint32_t kFlagMask = 0xFFFF0000u;

// stage row `id` as the wave's LDS query
template <bool kBf16>
__device__ __forceinline__ void stage_query(const HnswBuildArgs &a, float4 *qs, uint32_t id, int lane) {
  const char *src = row_base<kBf16>(a.rows, id, a.row_stride_f);
  for (uint32_t i = lane; i < a.chunks * 4; i += kWave) qs[i] = row_piece<kBf16>(src, i);   // bf16 rows widen here
}

// getNeighborsByHeuristic2's inner test for one candidate c (already staged in qs): is any kept
// node closer to c than c is to the base point?  kept ids live in LDS.
template <bool kL2, bool kBf16>
__device__ __forceinline__ bool dominated(const HnswBuildArgs &a, const float4 *qs, const uint32_t *kept, uint32_t nkept,
                                          float dist_to_base, int lane) {
  const int j = lane & 3, rq = lane >> 2;
  for (uint32_t base = 0; base < nkept; base += kRowsPerWave) {
    const uint32_t i = base + rq;
    const bool valid = i < nkept;
    const uint32_t sid = kept[valid ? i : 0];
    const float d = quad_row_distance<kL2, kBf16>(row_base<kBf16>(a.rows, sid, a.row_stride_f), qs, a.chunks, j);
    if (__ballot(valid && d < dist_to_base) != 0) return true;     // hnswalg.h:583-586
  }
  return false;
}
}  // namespace

// ---- selection for the new points --------------------------------------------------------------
// in : cand_id/cand_dist [n_new][ld] ascending by distance (hnsw_search_kernel, ids as u64), cand_n
// out: links0[first_id + p] = the selected neighbours; sel_* = the same, for the reverse links
template <bool kL2, bool kBf16>
__global__ __launch_bounds__(256) void hnsw_select_kernel(HnswBuildArgs a) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wave_f4 = (size_t)a.chunks * 4 + (a.max_keep + 3) / 4;
  float4 *qs = lds4 + wave * per_wave_f4;
  uint32_t *kept = reinterpret_cast<uint32_t *>(qs + a.chunks * 4);

  const uint32_t p = blockIdx.x * 4 + wave;
  if (p >= a.n_new) return;
  const uint32_t n = a.cand_n[p] < a.cand_ld ? a.cand_n[p] : a.cand_ld;
  const uint64_t *cid = a.cand_id + (size_t)p * a.cand_ld;
  const float *cd = a.cand_dist + (size_t)p * a.cand_ld;
  uint32_t *sel_id = a.sel_id + (size_t)p * a.max_keep;
  float *sel_d = a.sel_dist + (size_t)p * a.max_keep;

  uint32_t nk = 0;
  if (n < a.max_keep) {                                    // hnswalg.h:555-557: fewer than M -> keep all
    for (uint32_t i = lane; i < n; i += kWave) { kept[i] = (uint32_t)cid[i]; sel_d[i] = cd[i]; }
    nk = n;
  } else {
    for (uint32_t ci = 0; ci < n && nk < a.max_keep; ++ci) {
      const uint32_t c = (uint32_t)cid[ci];
      const float dq = cd[ci];
      bool good = true;
      if (nk) {
        stage_query<kBf16>(a, qs, c, lane);
        good = !dominated<kL2, kBf16>(a, qs, kept, nk, dq, lane);
      }
      if (good) {
        if (lane == 0) { kept[nk] = c; sel_d[nk] = dq; }
        ++nk;
      }
    }
  }
  uint32_t *ll = a.links0 + (size_t)(a.first_id + p) * a.l0_stride;
  for (uint32_t i = lane; i < nk; i += kWave) {
    const uint32_t v = kept[i];
    ll[1 + i] = v;
    sel_id[i] = v;
  }
  if (lane == 0) {
    ll[0] = (ll[0] & kFlagMask) | nk;
    a.sel_n[p] = nk;
  }
}

// ---- reverse links ---------------------------------------------------------------------------------
// touched node t: s = node[t]; new points add_p[off[t]..off[t+1]) at distances add_d (ascending).
// Room left: append.  Otherwise the old list (distances to s computed here) and the new points go
// through the heuristic together with capacity maxM0 (hnswalg.h:706-738).
template <bool kL2, bool kBf16>
__global__ __launch_bounds__(256) void hnsw_relink_kernel(HnswBuildArgs a) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 3, rq = lane >> 2;
  constexpr uint32_t kCap = 256;                            // candidates per node (old list + additions)
  const size_t per_wave_f4 = (size_t)a.chunks * 4 + (kCap * 4 + a.max_keep + 3) / 4;
  float4 *qs = lds4 + wave * per_wave_f4;
  float *c_d = reinterpret_cast<float *>(qs + a.chunks * 4);
  uint32_t *c_id = reinterpret_cast<uint32_t *>(c_d + kCap);
  float *s_d = reinterpret_cast<float *>(c_id + kCap);      // sorted copies
  uint32_t *s_id = reinterpret_cast<uint32_t *>(s_d + kCap);
  uint32_t *kept = s_id + kCap;

  const uint32_t t = blockIdx.x * 4 + wave;
  if (t >= (a.counts ? a.counts[0] : a.n_touched)) return;
  const uint32_t s = a.node[t];
  uint32_t *ll = a.links0 + (size_t)s * a.l0_stride;
  const uint32_t w0 = ll[0];
  const uint32_t cnt = w0 & 0xFFFFu;
  const uint32_t a0 = a.off[t];
  uint32_t nadd = a.off[t + 1] - a0;
  if (cnt + nadd <= a.max_keep) {                           // hnswalg.h:688-704
    for (uint32_t i = lane; i < nadd; i += kWave) ll[1 + cnt + i] = a.add_p[a0 + i];
    if (lane == 0) ll[0] = (w0 & kFlagMask) | (cnt + nadd);
    return;
  }
  if (cnt + nadd > kCap) nadd = kCap - cnt;                 // additions are sorted: the farthest are dropped

  // candidates: old neighbours with their distance to s, then the new points
  stage_query<kBf16>(a, qs, s, lane);
  for (uint32_t base = 0; base < cnt; base += kRowsPerWave) {
    const uint32_t i = base + rq;
    const bool valid = i < cnt;
    const uint32_t e = ll[1 + (valid ? i : 0)];
    const float d = quad_row_distance<kL2, kBf16>(row_base<kBf16>(a.rows, e, a.row_stride_f), qs, a.chunks, j);
    if (valid && j == 0) { c_d[i] = d; c_id[i] = e; }
  }
  for (uint32_t i = lane; i < nadd; i += kWave) { c_d[cnt + i] = a.add_d[a0 + i]; c_id[cnt + i] = a.add_p[a0 + i]; }
  const uint32_t total = cnt + nadd;
  // ascending by (distance, id): rank by counting
  for (uint32_t i = lane; i < total; i += kWave) {
    const float di = c_d[i];
    const uint32_t ii = c_id[i];
    uint32_t rank = 0;
    for (uint32_t o = 0; o < total; ++o) {
      const float dO = c_d[o];
      const uint32_t io = c_id[o];
      rank += (dO < di || (dO == di && io < ii)) ? 1u : 0u;
    }
    s_d[rank] = di;
    s_id[rank] = ii;
  }
  uint32_t nk = 0;
  for (uint32_t ci = 0; ci < total && nk < a.max_keep; ++ci) {
    const uint32_t c = s_id[ci];
    const float dq = s_d[ci];
    bool good = true;
    if (nk) {
      stage_query<kBf16>(a, qs, c, lane);
      good = !dominated<kL2, kBf16>(a, qs, kept, nk, dq, lane);
    }
    if (good) {
      if (lane == 0) kept[nk] = c;
      ++nk;
    }
  }
  for (uint32_t i = lane; i < nk; i += kWave) ll[1 + i] = kept[i];
  if (lane == 0) ll[0] = (w0 & kFlagMask) | nk;
}

// ---- grouping the (new point -> selected neighbour) pairs by neighbour, on the device ---------------
namespace {
__device__ __forceinline__ uint32_t f32_order_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void hnsw_pairs_kernel(const uint32_t *sel_id, const float *sel_dist, const uint32_t *sel_n, uint32_t n_new,
                                  uint32_t m, uint32_t first, uint64_t *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new * m) return;
  const uint32_t p = i / m, t = i % m;
  const bool valid = t < sel_n[p];
  keys[i] = valid ? (((uint64_t)sel_id[i] << 32) | f32_order_key(sel_dist[i])) : ~0ull;
  vals[i] = first + p;
}
__global__ void hnsw_heads_kernel(const uint64_t *keys, uint32_t n, uint32_t *flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  flags[i] = (k != ~0ull && (i == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(k >> 32))) ? 1u : 0u;
}
// node[t], off[t] for the heads; add_d for every pair; counts = {touched nodes, pairs}
__global__ void hnsw_csr_kernel(const uint64_t *keys, const uint32_t *flags, const uint32_t *pos, uint32_t n,
                                uint32_t *node, uint32_t *off, float *add_d, uint32_t *counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  const bool valid = k != ~0ull;
  if (i == 0 && !valid) { counts[0] = 0; counts[1] = 0; off[0] = 0; }
  if (!valid) return;
  add_d[i] = f32_from_order_key((uint32_t)k);
  if (flags[i]) { node[pos[i]] = (uint32_t)(k >> 32); off[pos[i]] = i; }
  if (i == n - 1 || keys[i + 1] == ~0ull) {        // the last pair
    const uint32_t T = pos[i] + flags[i];
    counts[0] = T;
    counts[1] = i + 1;
    off[T] = i + 1;
  }
}
// the lists a batch wrote, compacted for the host: rows 0..n_new-1 = the new points, then the touched nodes
__global__ void hnsw_gather_lists_kernel(uint32_t *dst, const uint32_t *links0, uint32_t stride, uint32_t first,
                                         uint32_t n_new, const uint32_t *node, const uint32_t *counts) {
  const uint64_t rows = (uint64_t)n_new + counts[0];
  const uint64_t total = rows * stride;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(t / stride), w = (uint32_t)(t % stride);
    const uint32_t id = r < n_new ? first + r : node[r - n_new];
    dst[t] = links0[(size_t)id * stride + w];
  }
}
}  // namespace

// bf16 rows -> f32 query block for the beam search (queries are always f32)
__global__ void hnsw_widen_rows_kernel(const uint16_t *rows, uint32_t stride_e, uint32_t first, uint32_t n, float *out) {
  const uint64_t total = (uint64_t)n * stride_e;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x)
    out[t] = __uint_as_float((uint32_t)rows[(size_t)first * stride_e + t] << 16);
}
hipError_t launch_hnsw_widen_rows(const void *rows, uint32_t stride_e, uint32_t first, uint32_t n, float *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hnsw_widen_rows_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t *>(rows), stride_e, first,
                     n, out);
  return hipGetLastError();
}

// ---- sort and scan of a batch's (target node, distance) pairs ---------------------------------------------------------
// At most a few hundred thousand pairs per batch: a stable LSD radix sort, 8 bits per pass, one wave per chunk of the
// input (histogram -> one-block exclusive scan of the chunk histograms, digit-major -> stable scatter in input order),
// and a two-level exclusive scan of the head flags.
namespace {
constexpr uint32_t kSortChunks = 128;     // blocks (= waves) per pass
constexpr uint32_t kScanBlock = 1024;     // flags per block of the scan

__global__ __launch_bounds__(64) void radix_hist_kernel(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t *hist) {
  __shared__ uint32_t h[256];
  const uint32_t lane = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  for (uint32_t d = lane; d < 256; d += kWave) h[d] = 0;
  __syncthreads();
  const uint32_t chunk = (n + nb - 1) / nb, lo = b * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (uint32_t i = lo + lane; i < hi; i += kWave) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  for (uint32_t d = lane; d < 256; d += kWave) hist[d * nb + b] = h[d];   // digit-major: a scan over it is the scatter order
}
// exclusive scan of `total` counters by one block (total = 256 digits x chunks, or the block sums of the flag scan)
__global__ __launch_bounds__(256) void scan_one_block_kernel(uint32_t *v, uint32_t total) {
  __shared__ uint32_t part[256];
  const uint32_t t = threadIdx.x, per = (total + 255) / 256, lo = t * per, hi = lo + per < total ? lo + per : total;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += v[i];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { const uint32_t c = part[i]; part[i] = run; run += c; }
  }
  __syncthreads();
  uint32_t run = part[t];
  for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = v[i]; v[i] = run; run += c; }
}
__global__ __launch_bounds__(64) void radix_scatter_kernel(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout,
                                                           uint32_t n, uint32_t shift, const uint32_t *hist) {
  __shared__ uint32_t off[256];
  const uint32_t lane = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  for (uint32_t d = lane; d < 256; d += kWave) off[d] = hist[d * nb + b];
  __syncthreads();
  const uint32_t chunk = (n + nb - 1) / nb, lo = b * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (uint32_t base = lo; base < hi; base += kWave) {
    const uint32_t i = base + lane;
    const bool valid = i < hi;
    const uint64_t key = valid ? kin[i] : 0;
    const uint32_t val = valid ? vin[i] : 0;
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    // the lanes of this round with the same digit; a lane's place among them = its rank in input order (stability)
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const uint64_t m = __ballot(one);
      peers &= one ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), cnt = (uint32_t)__popcll(peers);
    uint32_t dst = 0;
    if (valid) dst = off[d] + rank;
    __syncthreads();
    if (valid) {
      kout[dst] = key;
      vout[dst] = val;
      if (rank == cnt - 1) off[d] += cnt;      // one lane per digit moves the digit's cursor on
    }
    __syncthreads();
  }
}
// flags -> exclusive positions: per block of 1024, then the block sums, then the offsets added back
__global__ __launch_bounds__(256) void scan_blocks_kernel(const uint32_t *flags, uint32_t n, uint32_t *pos, uint32_t *sums) {
  __shared__ uint32_t part[256];
  const uint32_t t = threadIdx.x, lo = blockIdx.x * kScanBlock + t * 4;
  uint32_t f[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { f[u] = lo + u < n ? flags[lo + u] : 0; sum += f[u]; }
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { const uint32_t c = part[i]; part[i] = run; run += c; }
    sums[blockIdx.x] = run;
  }
  __syncthreads();
  uint32_t run = part[t];
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (lo + u < n) pos[lo + u] = run; run += f[u]; }
}
__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *pos, uint32_t n, const uint32_t *sums) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] += sums[i / kScanBlock];
}
}  // namespace

// scratch of launch_hnsw_group: the sort's second buffer (keys + values), the chunk histograms, the scan's block sums
size_t hnsw_group_tmp_bytes(uint32_t n_pairs) {
  const size_t n = n_pairs;
  return ((n * 8 + 255) & ~(size_t)255) + ((n * 4 + 255) & ~(size_t)255) + (size_t)256 * kSortChunks * 4 +
         (((n + kScanBlock - 1) / kScanBlock) * 4 + 255 & ~(size_t)255) + 256;
}

// sel_* -> CSR {node, off, add_p, add_d, counts}; everything on stream s, no host round trip
hipError_t launch_hnsw_group(const HnswGroupArgs &g, hipStream_t s) {
  const uint32_t n = g.n_new * g.m;
  if (n == 0) return hipSuccess;
  if (g.tmp_bytes < hnsw_group_tmp_bytes(n)) return hipErrorInvalidValue;
  const uint32_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(hnsw_pairs_kernel, dim3(blocks), dim3(256), 0, s, g.sel_id, g.sel_dist, g.sel_n, g.n_new, g.m,
                     g.first_id, g.keys_a, g.vals_a);
  // (keys_a, vals_a) -> (keys_b, add_p), ascending by (node, distance), equal keys in input order: eight passes that
  // alternate between the input buffers and the scratch pair and end in the output pair
  char *tp = static_cast<char *>(g.tmp);
  uint64_t *keys_t = reinterpret_cast<uint64_t *>(tp);
  tp += ((size_t)n * 8 + 255) & ~(size_t)255;
  uint32_t *vals_t = reinterpret_cast<uint32_t *>(tp);
  tp += ((size_t)n * 4 + 255) & ~(size_t)255;
  uint32_t *hist = reinterpret_cast<uint32_t *>(tp);
  tp += (size_t)256 * kSortChunks * 4;
  uint32_t *sums = reinterpret_cast<uint32_t *>(tp);
  for (uint32_t pass = 0; pass < 8; ++pass) {
    const uint64_t *kin = (pass & 1) ? keys_t : g.keys_a;
    const uint32_t *vin = (pass & 1) ? vals_t : g.vals_a;
    uint64_t *kout = pass == 7 ? g.keys_b : (pass & 1) ? g.keys_a : keys_t;
    uint32_t *vout = pass == 7 ? g.add_p : (pass & 1) ? g.vals_a : vals_t;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(kSortChunks), dim3(64), 0, s, kin, n, pass * 8, hist);
    hipLaunchKernelGGL(scan_one_block_kernel, dim3(1), dim3(256), 0, s, hist, 256u * kSortChunks);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(kSortChunks), dim3(64), 0, s, kin, vin, kout, vout, n, pass * 8, hist);
  }
  hipLaunchKernelGGL(hnsw_heads_kernel, dim3(blocks), dim3(256), 0, s, g.keys_b, n, g.flags);
  const uint32_t sblocks = (n + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(sblocks), dim3(256), 0, s, g.flags, n, g.pos, sums);
  hipLaunchKernelGGL(scan_one_block_kernel, dim3(1), dim3(256), 0, s, sums, sblocks);
  hipLaunchKernelGGL(scan_add_kernel, dim3(blocks), dim3(256), 0, s, g.pos, n, sums);
  hipLaunchKernelGGL(hnsw_csr_kernel, dim3(blocks), dim3(256), 0, s, g.keys_b, g.flags, g.pos, n, g.node, g.off, g.add_d,
                     g.counts);
  return hipGetLastError();
}

hipError_t launch_hnsw_gather_lists(uint32_t *dst, const uint32_t *links0, uint32_t stride, uint32_t first, uint32_t n_new,
                                    const uint32_t *node, const uint32_t *counts, hipStream_t s) {
  hipLaunchKernelGGL(hnsw_gather_lists_kernel, dim3(2048), dim3(256), 0, s, dst, links0, stride, first, n_new, node, counts);
  return hipGetLastError();
}

size_t hnsw_build_lds_bytes(const HnswBuildArgs &a, bool relink) {
  const size_t per_wave_f4 = (size_t)a.chunks * 4 + ((relink ? 256 * 4 : 0) + a.max_keep + 3) / 4;
  return per_wave_f4 * 16 * 4;
}

hipError_t launch_hnsw_select(const HnswBuildArgs &a, bool l2, bool bf16, hipStream_t s) {
  const size_t lds = hnsw_build_lds_bytes(a, false);
  const void *fn = l2 ? (bf16 ? reinterpret_cast<const void *>(&hnsw_select_kernel<true, true>)
                              : reinterpret_cast<const void *>(&hnsw_select_kernel<true, false>))
                      : (bf16 ? reinterpret_cast<const void *>(&hnsw_select_kernel<false, true>)
                              : reinterpret_cast<const void *>(&hnsw_select_kernel<false, false>));
  hipError_t e = ensure_max_lds(fn);
  if (e != hipSuccess) return e;
  HnswBuildArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(fn, dim3((a.n_new + 3) / 4), dim3(256), params, lds, s);
}

hipError_t launch_hnsw_relink(const HnswBuildArgs &a, bool l2, bool bf16, hipStream_t s) {
  if (a.n_touched == 0) return hipSuccess;    // with a.counts: n_touched is the upper bound (n_new * M)
  const size_t lds = hnsw_build_lds_bytes(a, true);
  const void *fn = l2 ? (bf16 ? reinterpret_cast<const void *>(&hnsw_relink_kernel<true, true>)
                              : reinterpret_cast<const void *>(&hnsw_relink_kernel<true, false>))
                      : (bf16 ? reinterpret_cast<const void *>(&hnsw_relink_kernel<false, true>)
                              : reinterpret_cast<const void *>(&hnsw_relink_kernel<false, false>));
  hipError_t e = ensure_max_lds(fn);
  if (e != hipSuccess) return e;
  HnswBuildArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(fn, dim3((a.n_touched + 3) / 4), dim3(256), params, lds, s);
}

}  // namespace vk
