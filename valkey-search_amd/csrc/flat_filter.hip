// flat_filter.hip -- K4h: the candidate stage of the batched FLAT search on the f16 / bf16 matrix cores.
//
// The exact batched kernel (flat_gemm.hip) reproduces the reference's f32 arithmetic on the f32 MFMA pipe, which is
// 1/16 of the f16 rate: 36.7 ms per 256-query batch at 10M x 768, where ONE pass over the rows costs 4.9 ms of HBM
// time.  An exact answer does not need exact arithmetic for every row: this kernel computes an APPROXIMATE score
// (dot product; for L2 dot - |x|^2/2) of every (row, query) pair with v_mfma_f32_32x32x16_f16 (rows converted f32 ->
// f16 on the way into LDS, queries converted once) and keeps only the pairs that can still be among the query's k best:
//
//   approx(x, q) >= L_q - E_q(R_t)
//
// L_q is a lower bound of the query's k-th best EXACT score and E_q(R) bounds |approx - exact| (plus the reference's
// own rounding) rigorously for a row of norm <= R; R_t is the largest row norm of the 128-row tile the row lives in.
// The survivors -- a few hundred per query out of 10M -- go, WITH their approximate scores, to the fused re-rank
// (flat_rerank_kernel, flat_scan.hip): a second bound taken from those scores (the k-th largest score - E over a query's
// survivors bounds its k-th best exact score from the whole index, not from the sample) leaves about k + 1 of them, which
// get an exact distance from the quad kernel's arithmetic and are selected by (distance,label) -- so the answer is
// BIT-IDENTICAL to the exact path's; only the work differs.
//
// Where L_q comes from (r03; r02 ran the exact kernel over the first rows, a filter pass over a larger prefix and a
// re-rank of that -- eleven launches, and a bound that depended on the order the index was loaded in): one pass of THIS
// kernel in "sample mode" over every s-th row tile of the index writes, per (group of 64 rows, query), the group's best
// approximate score minus its margin -- a lower bound of the exact score of SOME row of the group.  The k-th largest of
// a query's group bounds (flat_bound_select_kernel) is reached by k distinct rows, hence bounds the k-th best exact
// score from below.  No exact arithmetic is needed for the bound, and a strided sample sees every region of an index
// that was loaded cluster by cluster or in time order (the reference's answer does not depend on row order:
// bruteforce.h:116-145).
//
// Error bound.  x^ = rne_f16(x), q^ = rne_f16(q): |x^_i - x_i| <= 2^-11 |x_i| + 2^-25 (the second term covers f16
// subnormals), same for q.  Products of f16 values are exact in f32; the MFMA accumulates in f32, allowed here 4 ulp
// per accumulated term (D * 2^-22 relative to sum |x_i q_i|), far more than an IEEE chain needs.  With
// sum |x_i q_i| <= |x| |q| (Cauchy-Schwarz; tight exactly for the near neighbours that matter):
//   |approx - dot_real| <= |x||q| (2^-11 + 2^-11 + 2^-22 + D 2^-22) + 2^-25 sqrt(D) (|x| + |q|) (1 + 2^-11)
// and the reference's own f32 result differs from dot_real by at most (D/16 + 5) 2^-24 |x||q|, its 1 - dot by 2^-24
// max(1, |dist|).  As a polynomial in the row norm R with per-query coefficients (flat_qprep_kernel): E_q(R) = c2 R^2 +
// c1 R + c0 (c2 = 0 for the inner-product space); the margin is applied once for the witness rows behind L_q and once
// for the row at the gate.  Because R is per TILE, one long row in an un-normalised index widens the gate of its own
// 128 rows only (r02: of every row).
//
// Nothing is dropped silently and no single query can take the batch down with it (r02: one overflowing list or one
// value outside f16 anywhere in the index sent all 256 queries to the 36.7 ms exact kernel):
//   * a tile holding a value the f16 pipe cannot carry (non-finite, beyond 32768, a half norm beyond f16) has
//     R_t = +inf: every pair of that tile survives and the exact re-rank settles it;
//   * a query with more survivors than its private list (duplicates of one vector by the ten thousand) continues in
//     spill chunks handed out from a shared pool; only when that is exhausted -- or the query itself cannot go through
//     f16 -- is the query (alone) marked for the exact redo pass that FlatIndex enqueues behind.
//
// Data movement per 128-row tile and block (eight waves with three different jobs, see the kernel):
//   rows     393 KB f32 from HBM, once -> converted -> f16 in LDS (two stages of 18 KB, 144-B row stride: conflict-free
//            ds_read_b128) -> A operands of the four multiplying waves
//   queries  the batch's f16 copy in MFMA fragment order (flat_qprep_kernel, 384 KB at 256 x 768) comes from L2, one
//            32 KB stage at a time, through LDS -> B operands; in the final pass by DMA (buffer_load ... lds) into a ring
//            of three stages, no registers and no LDS store instructions in between (kBDma)
// so HBM traffic is the row bytes and L2 traffic twice that (PMC: TCC misses 30.8 GB, hits 30.8 GB per launch).
// bf16 rows in the inner-product space (kBfMma, kDma): rows and queries stay bf16, the bf16 matrix-core instruction
// multiplies (query rounding 2^-8 in the margin), and the ROWS go HBM -> LDS by DMA (swizzled 128-B rows, ring of five).
// Roofline: HBM (30.72 GB per launch at 10M x 768 f32).  The matrix cores: 3.9 PFLOP of f16 per launch; their stream alone
// (no operands fetched, no gate) takes 2.3-2.4 ms of a 5.1-5.2 ms launch at the clock the chip sustains under it --
// scripts/filter_ablate.py and DESIGN.md section 5 have the whole ablation matrix (VK_FILTER_ABLATE below).
#include <stdlib.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace vk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// experiment-only arguments (kernels.hpp): constants in the product build, so the variants they select do not exist there
#ifdef VK_EXPERIMENTS
__device__ __forceinline__ uint32_t exp_ablate(const FlatFilterArgs &a) { return a.ablate; }
__device__ __forceinline__ uint32_t exp_prio(const FlatFilterArgs &a) { return a.prio; }
__device__ __forceinline__ unsigned long long *exp_dbg(const FlatFilterArgs &a) { return a.dbg; }
#else
__device__ __forceinline__ uint32_t exp_ablate(const FlatFilterArgs &) { return 0u; }
__device__ __forceinline__ uint32_t exp_prio(const FlatFilterArgs &) { return 0u; }
__device__ __forceinline__ unsigned long long *exp_dbg(const FlatFilterArgs &) { return nullptr; }
#endif

namespace {
constexpr int kFTileRows = 128;
constexpr int kFStageK = 64;                 // k per pipeline stage: 4 MFMA K-steps of 16
constexpr int kFAStride = 72;                // halfs per staged row: 64 + 8 pad = 144 B (conflict-free b128 reads)
}  // namespace

// ---- row statistics: the largest row norm per 128-row tile (and of the index) over rows [lo, hi) ------------------------
// tile_norm[t] = max over the tile's rows of |x| (the norm, f32 bits, rounded up), or +inf when the tile holds a value the f16
// pipe cannot carry (non-finite, |x_i| > 32768, for L2 a half norm beyond f16): the filter lets every pair of such a
// tile through to the exact re-rank.  stats[0] / stats[1] = the same maxima over the whole index (|x|^2, |x_i|; reported,
// not used by the gate), stats[2] = number of tiles flagged +inf (FlatIndex keeps an index that is mostly such tiles
// off this path).  All of them only ever grow (atomicMax on the bit patterns of non-negative floats), which keeps them
// valid bounds when rows are overwritten or removed.
// hn16 (optional, L2 indexes): per row half its squared norm, split into two f16 (hi | lo << 16) -- the extra K-step
// that turns the filter's dot product into dot - |x|^2 / 2.
__global__ __launch_bounds__(256) void row_stats_kernel(const void *rows, uint32_t bf16, uint32_t l2, uint32_t stride_e, uint32_t chunks,
                                                        uint32_t lo, uint32_t hi, uint32_t *stats, uint32_t *tile_norm, uint32_t *hn16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 3, rq = lane >> 2;
  const uint32_t total_waves = gridDim.x * 4, n_tiles = (hi - lo + kRowsPerWave - 1) / kRowsPerWave;
  float best_n2 = 0.f, best_abs = 0.f;
  for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += total_waves) {   // (lo is a multiple of 128)
    const uint32_t row = lo + tile * kRowsPerWave + rq;
    const uint32_t lrow = row < hi ? row : hi - 1;
    float n2 = 0.f, mx = 0.f;
    bool bad = false;
    for (uint32_t c = 0; c < chunks; ++c) {
      const float4 x = bf16 ? row_piece<true>(row_base<true>(rows, lrow, stride_e), c * 4 + j)
                            : row_piece<false>(row_base<false>(rows, lrow, stride_e), c * 4 + j);
      n2 = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, n2))));
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
      bad = bad || !(x.x - x.x == 0.f) || !(x.y - x.y == 0.f) || !(x.z - x.z == 0.f) || !(x.w - x.w == 0.f);
    }
    n2 += dpp_quad_xor1(n2);
    n2 += dpp_quad_xor2(n2);
    if (bad || !(n2 - n2 == 0.f)) mx = __builtin_inff();
    if (hn16 != nullptr && row < hi && j == 0) {
      const float hn = 0.5f * n2;
      const _Float16 h0 = (_Float16)hn;
      const _Float16 h1 = (_Float16)(hn - (float)h0);
      hn16[row] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    }
    // the 16 rows of this step lie in one 128-row tile
    const float g_n2 = wave_max_f32(n2) * 1.0001f;   // (rounding of the sum itself: far below 1e-4)
    const float g_abs = wave_max_f32(mx);
    const bool g_bad = !(g_abs <= 32768.f) || !(g_n2 - g_n2 == 0.f) || (l2 && !(0.5f * g_n2 <= 60000.f));
    if (lane == 0) {
      const uint32_t v = g_bad ? 0x7F800000u : __float_as_uint(sqrtf(g_n2) * 1.0001f);   // the NORM, rounded up
      const uint32_t old = atomicMax(&tile_norm[(lo + tile * kRowsPerWave) / 128u], v);
      if (g_bad && old != 0x7F800000u) atomicAdd(&stats[2], 1u);
    }
    best_n2 = fmaxf(best_n2, g_n2);
    best_abs = fmaxf(best_abs, g_abs);
  }
  if (lane == 0) {
    atomicMax(&stats[0], __float_as_uint(best_n2));
    atomicMax(&stats[1], __float_as_uint(best_abs));
  }
}

// stats[3] = the norm cap of the sample's witnesses: the upper edge of the smallest tile-norm bin (exponent + 3 mantissa
// bits: 9 % wide) that 97 % of the finite tiles stay below.  A robust "largest ordinary norm": one row of norm 1e6 in a
// unit-norm index moves the global maximum by six orders of magnitude and this not at all.  One block.
__global__ __launch_bounds__(1024) void tile_cap_kernel(const uint32_t *tile_norm, uint32_t n_tiles, uint32_t *stats) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  hist[tid] = hist[tid + 1024] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n_tiles; i += 1024) {
    const uint32_t v = tile_norm[i];
    if (v < 0x7F800000u) atomicAdd(&hist[v >> 20], 1u);
  }
  __syncthreads();
  part[tid] = hist[2 * tid] + hist[2 * tid + 1];
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (uint32_t i = 0; i < 1024; ++i) total += part[i];
    const uint32_t want = total - total / 32;                     // 97 % of the finite tiles
    uint32_t acc = 0, bin = 2047;
    for (uint32_t b = 0; b < 2048 && total != 0; ++b) {
      acc += hist[b];
      if (acc >= want) { bin = b; break; }
    }
    stats[3] = total == 0 ? 0u : (bin >= 2039u ? 0x7F7FFFFFu : ((bin + 1u) << 20));
  }
}

hipError_t launch_row_stats(const void *rows, bool bf16, bool l2, uint32_t stride_e, uint32_t lo, uint32_t hi, uint32_t n_tiles,
                            uint32_t *stats, uint32_t *tile_norm, uint32_t *hn16, hipStream_t s) {
  lo &= ~127u;                                            // whole tiles: a step of 16 rows never straddles two of them
  if (hi > lo) {
    const uint32_t tiles = (hi - lo + kRowsPerWave - 1) / kRowsPerWave;
    const uint32_t blocks = std::min<uint32_t>((tiles + 3) / 4, 2048);
    hipLaunchKernelGGL(row_stats_kernel, dim3(blocks), dim3(256), 0, s, rows, bf16 ? 1u : 0u, l2 ? 1u : 0u, stride_e, stride_e / 16, lo, hi,
                       stats, tile_norm, hn16);
  }
  hipLaunchKernelGGL(tile_cap_kernel, dim3(1), dim3(1024), 0, s, tile_norm, n_tiles, stats);
  return hipGetLastError();
}

// ---- query preparation -------------------------------------------------------------------------------------------------
// One wave per query column of the (padded) batch: f16 copy in MFMA fragment order, the column's error polynomial, and
// the reset of everything the passes of this batch count in (survivor counts, spill chunks, hand-over flags).
// Fragment order: tile jt = j / 32 of 32 queries, K-step ks of 16 elements, lane l = g * 32 + (j % 32) holds elements
// ks*16 + g*8 + 0..7 -- the B operand of v_mfma_f32_32x32x16_f16 as one 16-byte load per lane.
__global__ __launch_bounds__(64) void flat_qprep_kernel(FlatFilterArgs a) {
  const uint32_t j = blockIdx.x, lane = threadIdx.x;
  const uint32_t q = j < a.nq ? j : a.nq - 1;          // padding columns replicate the last query (their gate never opens)
  const float *src = a.queries + (size_t)q * a.q_stride_f;
  const uint32_t ks_n = a.row_stride_f / 16, jt = j >> 5, jj = j & 31;
  float n2 = 0.f, mx = 0.f;
  bool bad = false;
  for (uint32_t k8 = lane; k8 < a.row_stride_f / 8; k8 += kWave) {
    const float4 u = reinterpret_cast<const float4 *>(src)[k8 * 2], v = reinterpret_cast<const float4 *>(src)[k8 * 2 + 1];
    const float e[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      n2 = fmaf(e[t], e[t], n2);
      mx = fmaxf(mx, fabsf(e[t]));
      bad = bad || !(e[t] - e[t] == 0.f);
    }
  }
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) {
    n2 += __shfl_xor(n2, m);
    mx = fmaxf(mx, __shfl_xor(mx, m));
    bad = bad || __shfl_xor((int)bad, m);
  }
  // a query that cannot go through f16 (values beyond its range, non-finite): its products could be NaN, which passes
  // every gate -- the column gets ZERO fragments and a closed gate, and the query is handed to the exact pass, alone
  const bool f16_ok = !bad && mx <= 32768.f;
  for (uint32_t k8 = lane; k8 < a.row_stride_f / 8; k8 += kWave) {
    const float4 u = reinterpret_cast<const float4 *>(src)[k8 * 2], v = reinterpret_cast<const float4 *>(src)[k8 * 2 + 1];
    const float e[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    f16x8 h;
    if (a.qbf16) {   // bf16 fragments (round to nearest even) for the bf16 matrix-core path over bf16 rows
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t u = __float_as_uint(e[t]);
        h[t] = __builtin_bit_cast(_Float16, (uint16_t)(f16_ok ? (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16 : 0u));
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) h[t] = f16_ok ? (_Float16)e[t] : (_Float16)0.f;   // round to nearest even
    }
    const uint32_t ks = k8 >> 1, g = k8 & 1;
    reinterpret_cast<f16x8 *>(a.q16)[((size_t)(jt * ks_n + ks) * kWave + g * 32 + jj)] = h;
  }
  if (j < a.nq && lane < kSpillPerQuery) a.qchunk[(size_t)j * kSpillPerQuery + lane] = 0u;
  if (lane != 0) return;
  if (j == 0) { *a.spill_next = 0u; *a.redo_cnt = 0u; }
  float4 co = make_float4(0.f, 0.f, 0.f, 1.f);          // padding column: closed
  if (j < a.nq) {
    a.cand_cnt[j] = 0u;
    if (a.done_cnt) a.done_cnt[j] = 0u;
    if (a.rerank_cnt) a.rerank_cnt[j] = 0u;
    const float qn = sqrtf(n2) * 1.0001f;
    const float D = (float)a.row_stride_f;
    // see the header: relative part, absolute (subnormal) part, the reference's own rounding, 1 - dot (2^-23 (1 + R |q|)
    // covers both sides' 2^-24 max(1, |dist|)); everything rounded up by 1.001
    // (bf16 rows convert to f16 exactly; on the bf16 matrix-core path the rows are not converted at all and the query is
    //  rounded to bf16 -- 8 significant bits, unit roundoff 2^-8 per element.  r03 had 2^-9 here: half the true bound,
    //  found by the margin audit on queries at bf16 rounding midpoints, tests/helpers/exp_margin_check.py: |approx - exact|
    //  reached 1.31 E)
    const float rel = (a.qbf16 ? 0x1p-8f : a.bf16 ? 0x1p-11f : 0x1p-10f) + 0x1p-22f + D * 0x1p-22f + (D / 16.f + 5.f) * 0x1p-24f;
    const float sub = 0x1.01p-25f * sqrtf(D);
    float c2 = 0.f, c1 = (qn * rel + sub + 0x1p-23f * qn) * 1.001f, c0 = (sub * qn + 0x1p-23f) * 1.001f;
    if (a.l2) {
      // |x - q|^2 = 2 (|x|^2/2) + |q|^2 - 2 x.q: the kernel accumulates x.q - |x|^2/2 (half norms as one more K-step,
      // split in two f16: 2^-21 relative), so in accumulator space the margin is eps2 / 2 with eps2 = twice the dot
      // product's margin, the f32 rounding of both norms (D 2^-23 relative, generously), the split of the half norm, and
      // the reference's own rounding of its sum of squared differences ((D/16 + 6) 2^-24 of at most (R + |q|)^2):
      //   eps2 / 2 = E_ip(R) + (R^2 + |q|^2) al / 2 + (R + |q|)^2 be / 2
      const float al = D * 0x1p-23f + 0x1p-20f, be = (D / 16.f + 6.f) * 0x1p-23f;
      c2 = 0.5f * (al + be) * 1.001f;
      c1 = (c1 + be * qn) * 1.001f;
      c0 = (c0 + 0.5f * qn * qn * (al + be)) * 1.001f;
    }
    const bool live = f16_ok && (c0 - c0 == 0.f) && (c1 - c1 == 0.f);
    co = make_float4(c2, c1, c0, live ? 0.f : 1.f);
    a.ovf_q[j] = live ? 0u : 1u;
  }
  a.qcoef[j] = co;
  // the column's margin for the witnesses of the sample: rows of norm up to the cap (see FlatFilterArgs::norm_cap)
  const float Rc = __uint_as_float(*a.norm_cap);
  a.qwit[j] = fmaf(fmaf(co.x, Rc, co.y), Rc, co.z);
}

// ---- bound selection --------------------------------------------------------------------------------------------------
// qbound[q] = the k-th largest of the query's group bounds (sample pass), found by a binary descent over the
// order-preserving keys: one block per query, up to 64 values per thread in registers, one barrier per bit.
template <int kPer>
__global__ __launch_bounds__(256) void flat_bound_select_kernel(FlatBoundArgs a) {
  __shared__ uint32_t s_cnt[2][4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
  const float *v = a.smax + (size_t)q * a.smax_ld;
  uint32_t key[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const uint32_t i = tid + 256u * u;
    key[u] = i < a.groups ? desc_key(v[i]) : 0u;          // (key 0 = below every float: never counted)
  }
  uint32_t phase = 0;
  auto block_sum = [&](uint32_t mine) -> uint32_t {
    if (lane == 0) s_cnt[phase][wave] = mine;
    __syncthreads();
    const uint32_t t = s_cnt[phase][0] + s_cnt[phase][1] + s_cnt[phase][2] + s_cnt[phase][3];
    phase ^= 1;
    return t;
  };
  // T = the largest key with count(key >= T) >= k, i.e. the k-th largest key (0 if there are fewer than k values)
  // (the top 20 bits: a bound 2^-11 short of the k-th largest group bound is as valid and costs twelve barriers less)
  uint32_t T = 0;
  for (int bit = 31; bit >= 12; --bit) {
    const uint32_t cand = T | (1u << bit);
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) c += (uint32_t)__popcll(__ballot(key[u] >= cand));
    if (block_sum(c) >= a.k) T = cand;
  }
  if (tid == 0) {
    const uint32_t u = (T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T;
    float b = __uint_as_float(u);
    if (T == 0 || !(b == b)) b = -__builtin_inff();       // fewer than k group bounds: no bound, the gate stays open
    a.qbound[q] = b;
  }
}

hipError_t launch_flat_bound_select(const FlatBoundArgs &a, hipStream_t s) {
  if (a.nq == 0) return hipSuccess;
  if (a.groups > kFilterMaxGroups) return hipErrorInvalidValue;
  if (a.groups <= 256 * 16) hipLaunchKernelGGL((flat_bound_select_kernel<16>), dim3(a.nq), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((flat_bound_select_kernel<64>), dim3(a.nq), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---- the early pass's harvest ------------------------------------------------------------------------------------------
// One wave per query: the first 64 kPer survivors of its list (any k distinct rows are witnesses; an early pass leaves a query
// a few hundred: kPer = 8 for k <= 64, 32 beyond), lo = score - margin at the row's tile norm, the k-th largest lo to 20 bits
// from below -- and the query's bound is raised to it.  The gate
// (gate_thr) charges the rounding of its own subtractions, so none is charged here.
template <bool kL2, int kPer>
__global__ __launch_bounds__(64) void flat_bound_tighten_kernel(FlatTightenArgs a) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  const uint32_t c_raw = a.cand_cnt[q];
  uint32_t n = c_raw < a.cap ? c_raw : a.cap;
  if (n > (uint32_t)kPer * kWave) n = (uint32_t)kPer * kWave;
  if (n < a.k) return;
  const float4 co = a.qcoef[q];
  if (co.w != 0.f) return;   // (a closed column: handed to the exact pass)
  uint32_t rowv[kPer];
  float valv[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const uint32_t i = (uint32_t)u * kWave + lane, ic = i < n ? i : n - 1;
    rowv[u] = a.cand_row[(size_t)q * a.cap + ic];
    valv[u] = a.cand_val[(size_t)q * a.cap + ic];
  }
  uint32_t key[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const float lo = valv[u] - filter_margin<kL2>(co.x, co.y, co.z, __uint_as_float(a.tile_norm[rowv[u] >> 7]));
    // (0 = below every float: never counted.  NaN, +-inf: a tile outside f16 -- no information)
    key[u] = ((uint32_t)u * kWave + lane < n && fabsf(lo) < __builtin_inff()) ? desc_key(lo) : 0u;
  }
  uint32_t T = 0;
  for (int bit = 31; bit >= 12; --bit) {
    const uint32_t cnd = T | (1u << bit);
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) c += (uint32_t)__popcll(__ballot(key[u] >= cnd));
    if (c >= a.k) T = cnd;
  }
  if (T == 0 || lane != 0) return;
  const float b = desc_key_float(T);
  if (b == b && b > a.qbound[q]) a.qbound[q] = b;
}

hipError_t launch_flat_bound_tighten(const FlatTightenArgs &a, hipStream_t s) {
  if (a.nq == 0) return hipSuccess;
  if (a.k <= 64) {
    if (a.l2) hipLaunchKernelGGL((flat_bound_tighten_kernel<true, 8>), dim3(a.nq), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((flat_bound_tighten_kernel<false, 8>), dim3(a.nq), dim3(64), 0, s, a);
  } else {
    if (a.l2) hipLaunchKernelGGL((flat_bound_tighten_kernel<true, 32>), dim3(a.nq), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((flat_bound_tighten_kernel<false, 32>), dim3(a.nq), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}

// ---- survivors, gate, stream position --------------------------------------------------------------------------------
// Survivors are collected per wave in LDS (a ring of 64 (query, row) entries) and written out 64 at a time: the global
// append is an atomicAdd that RETURNS the slot, a round trip of a microsecond or two -- paid per survivor it sat on the
// critical path of every row tile (some wave of the block nearly always had one, and the block's barrier waits for it).
struct SurvivorRing {
  // [64][3] LDS: (query, row slot, the pair's approximate score -- the re-rank's second bound works on these) side by side,
  // so that an entry is ONE LDS store of the consumer wave that found it (three dwords, stride 3: no bank conflicts)
  uint32_t *ent;
  uint32_t cnt;    // wave-uniform
};
constexpr int kRingWords = 3 * kWave;   // LDS words of one wave's ring
// A query's private list is full: the survivor goes to the query's spill chunks, handed out from a pool shared by the
// batch.  A chunk slot goes 0 (none) -> 1 (claimed: its chunk is being taken from the pool) -> id + 2, or kNoChunk when
// the pool is empty.  Exactly one thread claims a slot, so no chunk is ever lost; the others wait for the id.  Inside
// a wave the claims are made one (query, chunk) pair at a time by an elected lane BEFORE anybody waits, so a waiting
// lane only ever waits for another wave.  A survivor that finds no chunk (the query used all its slots, the pool is
// empty) is dropped -- and its query, alone, marked for the exact redo pass.
constexpr uint32_t kChunkClaimed = 1u, kNoChunk = 0xFFFFFFFFu;
__device__ __forceinline__ void ring_flush(const FlatFilterArgs &a, SurvivorRing &r, uint32_t lane) {
  bool spill = false;
  uint32_t q = 0, row = 0, j = 0;
  float val = 0.f;
  if (lane < r.cnt) {
    q = r.ent[lane * 3];
    row = r.ent[lane * 3 + 1];
    val = __uint_as_float(r.ent[lane * 3 + 2]);
    const uint32_t at = atomicAdd(&a.cand_cnt[q], 1u);
    if (at < a.cap) { a.cand_row[(size_t)q * a.cap + at] = row; a.cand_val[(size_t)q * a.cap + at] = val; }
    else { spill = true; j = at - a.cap; }
  }
  r.cnt = 0;
  if (__builtin_amdgcn_ballot_w64(spill) == 0) return;      // (the usual case)
  const uint32_t c = j / kSpillChunk;
  const bool has_slot = spill && c < kSpillPerQuery;
  uint32_t *slot = a.qchunk + (size_t)q * kSpillPerQuery + (has_slot ? c : 0u);
  bool need = has_slot && __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  for (uint64_t m = __builtin_amdgcn_ballot_w64(need); m != 0; m = __builtin_amdgcn_ballot_w64(need)) {
    const int leader = __builtin_ctzll(m);
    const uint32_t lq = (uint32_t)__builtin_amdgcn_readlane((int)q, leader), lc = (uint32_t)__builtin_amdgcn_readlane((int)c, leader);
    if ((int)lane == leader && atomicCAS(slot, 0u, kChunkClaimed) == 0u) {
      const uint32_t fresh = atomicAdd(a.spill_next, 1u);
      __hip_atomic_store(slot, fresh < a.n_chunks ? fresh + 2u : kNoChunk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    need = need && !(q == lq && c == lc);
  }
  uint32_t id = kNoChunk;
  if (has_slot) {
    while ((id = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == kChunkClaimed) __builtin_amdgcn_s_sleep(1);
  }
  if (spill) {
    if (id != kNoChunk) {
      a.spill[(size_t)(id - 2u) * kSpillChunk + j % kSpillChunk] = row;
      a.spill_val[(size_t)(id - 2u) * kSpillChunk + j % kSpillChunk] = val;
    } else __hip_atomic_store(a.ovf_q + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The column's gate for a tile whose largest row norm is R (norm_bits = R itself as f32 bits, rounded up by row_stats_kernel -- NOT
// its square; +inf = a tile that cannot go through
// f16: everything passes).  A pair stays unless approx < thr.
// (bound = +inf closes the column: a padding column, or a query handed to the exact pass)
template <bool kL2> struct GateCol { float c1, c0, bound; };
template <> struct GateCol<true> { float c2, c1, c0, bound; };
template <bool kL2> __device__ __forceinline__ float tile_margin(const GateCol<kL2> &c, float R) {
  if constexpr (kL2) return filter_margin<true>(c.c2, c.c1, c.c0, R);
  else return filter_margin<false>(0.f, c.c1, c.c0, R);
}
__device__ __forceinline__ float tile_norm(uint32_t r_bits) { return __uint_as_float(r_bits); }   // (row_stats rounded it up)
template <bool kL2> __device__ __forceinline__ float gate_thr(const GateCol<kL2> &c, uint32_t norm_bits) {
  const float R = tile_norm(norm_bits);
  // (2^-21 max(1, |bound|): the rounding of the subtractions that make the threshold out of bound and margin)
  float thr = (c.bound - tile_margin<kL2>(c, R)) - 0x1p-21f * fmaxf(1.f, fabsf(c.bound));
  if (!(thr == thr)) thr = -__builtin_inff();                       // (a tile beyond f16: R = +inf)
  return c.bound == __builtin_inff() ? __builtin_inff() : thr;
}

// Tile done: the gate.  Output register r of row tile rt is row rt*32 + (r&3) + 8*(r>>2) + 4*g, column li of the wave's
// query tile.  Almost every 32 x 32 block has no survivor: one max over the lane's 16 values, one ballot.  (The test is
// "not below", so that a NaN -- a tile or a query outside f16 -- passes.)
template <int kRt, bool kZero>
__device__ __forceinline__ void filter_gate(const FlatFilterArgs &a, f32x16 (&acc)[kRt], float thr, uint32_t tile_row0,
                                            uint32_t wave, uint32_t li, uint32_t g, SurvivorRing &ring, uint32_t lane) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef VK_EXPERIMENTS
  if (a.dump_scores != nullptr && tile_row0 < a.dump_rows) {   // margin audit: what this gate sees, for the test to judge
    const uint32_t q = wave * 32 + li;
    if (q < a.nq && q < a.dump_ld) {
#pragma unroll
      for (int rt = 0; rt < kRt; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = tile_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (row < a.dump_rows) a.dump_scores[(size_t)row * a.dump_ld + q] = acc[rt][r];
        }
      if (g == 0) a.dump_thr[(size_t)(tile_row0 / 128u) * a.dump_ld + q] = thr;
    }
  }
#endif
  // One vote for the wave's kRt blocks together: a vote is a trip from the vector to the scalar side (~60 cycles), and 96 %
  // of the blocks have no survivor.
  // (fmaxf drops NaNs; they only occur in a tile outside f16, whose thr is -inf: "not below" then holds for any m)
  float mx[kRt];
  bool any = false;
#pragma unroll
  for (int rt = 0; rt < kRt; ++rt) {
    mx[rt] = acc[rt][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx[rt] = fmaxf(mx[rt], acc[rt][r]);
    any = any || !(mx[rt] < thr);
  }
  if (__builtin_amdgcn_ballot_w64(any) == 0) {
    if constexpr (kZero) {
#pragma unroll
      for (int rt = 0; rt < kRt; ++rt) acc[rt] = zero;
    }
    return;
  }
#pragma unroll
  for (int rt = 0; rt < kRt; ++rt) {
    const float m = mx[rt];
    if (__builtin_amdgcn_ballot_w64(!(m < thr)) != 0) {   // (rare: a survivor somewhere in this 32 x 32 block)
      const uint32_t q = wave * 32 + li;
      // the lane's passing registers as a bit mask, then one round per remaining bit of the busiest lane (usually one)
      uint32_t mk = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) mk |= !(acc[rt][r] < thr) ? 1u << r : 0u;
      if (q >= a.nq) mk = 0;
      // (... or whose one passing register is not its maximum: a NaN, which fmaxf dropped)
      const bool multi = __builtin_amdgcn_ballot_w64(__builtin_popcount(mk) > 1 || (mk != 0 && m < thr)) != 0;
      while (__builtin_amdgcn_ballot_w64(mk != 0) != 0) {
        const uint32_t r = (uint32_t)__builtin_ctz(mk | 0x10000u);
        const uint32_t row = tile_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        bool pass = mk != 0 && row < a.n_rows;
        if (pass && a.allow_bits != nullptr) pass = allow_bit(a.allow_bits, a.allow_nbits, a.labels[row]);
        // the pair's approximate score, for the re-rank's second bound: a lane with ONE passing register holds it already
        // -- the maximum it just took -- and only a block in which some lane has several picks register r out of the
        // sixteen (a chain of selects: 25 us of the launch when every survivor paid for it)
        float score = m;
        if (multi) {
#pragma unroll
          for (int r2 = 0; r2 < 16; ++r2) score = r == (uint32_t)r2 ? acc[rt][r2] : score;
        }
        mk &= mk - 1;
        const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
        if (pm != 0) {
          const uint32_t n = (uint32_t)__popcll(pm);
          if (ring.cnt + n > kWave) ring_flush(a, ring, lane);
          if (pass) {
            const uint32_t at = ring.cnt + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
            ring.ent[at * 3] = q;
            ring.ent[at * 3 + 1] = row;
            ring.ent[at * 3 + 2] = __float_as_uint(score);
          }
          ring.cnt += n;
        }
      }
    }
    if constexpr (kZero) acc[rt] = zero;
  }
}

template <bool kBf16> __device__ __forceinline__ size_t sample_row(const FlatFilterArgs &a, uint32_t tile, uint32_t i);
// Sample mode: instead of gating, the lane's best approximate score over its rows of the sample tile minus the margin at
// the norm cap -- a lower bound of the exact score of one of those rows -- goes to smax[query][group].  Coarse groups (a
// large sample): the lane's 64 rows of the tile (half g of all four 32-row blocks), group = 2 * sample tile + g.  Fine
// groups (a small sample, where the k-th largest of few group bounds would be a poor bound): its 16 rows of each block,
// group = (4 * sample tile + block) * 2 + g.  Rows that are no witnesses arrive as NaN (fmaxf ignores them); with a
// filter only allowed rows count (a witness must be a row the search may return).
template <int kRt, bool kBf16>
__device__ __forceinline__ void sample_max(const FlatFilterArgs &a, const f32x16 (&acc)[kRt], float margin, bool closed, uint32_t tile,
                                           uint32_t q, uint32_t g) {
  float best = -__builtin_inff();
#pragma unroll
  for (int rt = 0; rt < kRt; ++rt) {
    float m = -__builtin_inff();
    if (a.allow_bits == nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[rt][r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t i = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (allow_bit(a.allow_bits, a.allow_nbits, a.labels[sample_row<kBf16>(a, tile, i)])) m = fmaxf(m, acc[rt][r]);
      }
    }
    if (a.smax_fine) {
      float lb = m - margin;
      if (closed || !(lb == lb)) lb = -__builtin_inff();
      if (q < a.nq) a.smax[(size_t)q * a.smax_ld + (tile * (uint32_t)kRt + rt) * 2u + g] = lb;
    }
    best = fmaxf(best, m);
  }
  if (!a.smax_fine) {
    float lb = best - margin;
    if (closed || !(lb == lb)) lb = -__builtin_inff();
    if (q < a.nq) a.smax[(size_t)q * a.smax_ld + tile * 2u + g] = lb;
  }
}

// position in a block's flattened (tile, stage) stream, advanced without divisions; it never moves past the last
// stage (prefetches behind the end re-read it and are not used)
struct FPos { uint32_t row0, st, left, step; };   // step: rows from one tile of the stream to the next (sample mode: a stride)
__device__ __forceinline__ void fpos_advance(FPos &p, uint32_t stages) {
  const bool go = p.left > 1;
  const bool wrap = go && p.st + 1 == stages;
  p.left -= p.left != 0 ? 1u : 0u;
  p.st = wrap ? 0u : p.st + (go ? 1u : 0u);
  p.row0 += wrap ? p.step : 0u;
}

// ---- the filter, wave-specialised ---------------------------------------------------------------------------------------
// Round 2's first version had eight identical waves (each its own query tile, B operands from L2 into registers, a
// share of the row loads); its cycle counters said that all eight multiply at the same time and then all eight issue
// their loads at the same time (72 wave-loads through one address path) -- the matrix pipes and the address path took
// turns, and the sum of the two was longer than the HBM time of the stage.  Here they belong to different waves:
//   waves 0-3  CONSUMERS, one per SIMD: wave w multiplies the 128 rows of the tile with query tiles 2w, 2w+1 (eight
//              32 x 32 accumulator tiles, 32 MFMAs per stage back to back).  A AND B operands come from LDS; the wave
//              issues no memory instruction at all
//   waves 4-5  ROW PRODUCERS: stream the rows, HBM -> registers (three stages in flight) -> f16 -> LDS
//   waves 6-7  QUERY PRODUCERS: the stage's B operands, L2 -> registers (two stages in flight) -> LDS
// so each SIMD holds one wave that keeps the matrix pipe busy and one that keeps the memory pipe busy, and a producer
// that waits for HBM holds up nobody's MFMAs.  One barrier per stage; LDS: A 2 x 18 KB, B 2 x 32 KB.
//
// Why rows and B operands have producers of their own: loads return in order (one vmcnt counter per wave), so a wave
// that fetches both waits for a B block (an L2 hit, needed soon) by waiting for the HBM rows requested before it; and
// with all four producers fetching both, the stage's 64 wave-loads were issued more slowly than by two and two (cycle
// counters: 1 900 against 1 270 cycles per stage -- the CU's one address path is the contended resource, 16 cycles per
// 64-lane x 16-byte load at best).  Why the B operands go through registers although global_load_lds could write them to
// LDS directly: a CU keeps only a few LDS-DMA pieces in flight -- eight per stage and wave took 1 350 cycles to ISSUE,
// more than the MFMAs of the stage.
//
// The producers' loads are ordinary loads and the compiler places the waits (it counts the loads issued after the
// one whose data is needed); what has to be kept away from it is LDS-DMA in the same
// wave (it then waits with vmcnt(0) everywhere) and __syncthreads() in the producers (below).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
// (pz, sample mode: bit u = the row of piece u comes from a tile beyond the norm cap and is poisoned on its way into LDS)
template <bool kBf16> struct WsRows { f32x4v v[16]; uint32_t hn, pz; };    // row producer thread: 128 rows x 64 k / 128 threads
template <> struct WsRows<true> { u32x4v v[8]; uint32_t hn, pz; };         // bf16 rows: 8 elements per 16-byte load, half as many loads
struct WsB { u32x4v v[16]; };                                          // query producer thread: 4 query tiles x 4 K-steps

// idx = t + 128 u: row = idx / 16 = t / 16 + 8 u, 4-element column t % 16 of the row's stage slice.  The tile / stage /
// u part of the address is uniform (a scalar base), the thread's part (voff, bytes) is computed once.
// The producers load through buffer descriptors: the wave-uniform part of an address (tile, stage, row group) lives in
// scalar registers -- descriptor base and scalar offset -- and the thread supplies one 32-bit offset computed once
// (buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen).  With 64-bit per-lane addresses the same 16 loads took half as
// long again to issue.  (0x00020000: the gfx9 raw-buffer format word; no bounds are wanted, the range is the maximum.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
template <bool kBf16, bool kL2>
__device__ __forceinline__ void ws_rows_load(WsRows<kBf16> &s, const FlatFilterArgs &a, uint32_t tile_row0, uint32_t st, uint32_t t, uint32_t voff) {
  if constexpr (kL2) {
    const uint32_t r = tile_row0 + t;
    s.hn = a.hn16[r < a.n_rows ? r : a.n_rows - 1];
  }
  constexpr size_t esz = kBf16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t r =
      ws_rsrc(static_cast<const char *>(a.rows) + ((size_t)tile_row0 * a.row_stride_f + (size_t)st * kFStageK) * esz);
  // (aux 2 = nt: the rows are read once -- they should not push the query block out of L2)
  if constexpr (kBf16) {
    // idx = t + 128 u: row = idx / 8 = t / 8 + 16 u, 8-element column t % 8 of the row's stage slice
    const uint32_t step = 16u * a.row_stride_f * (uint32_t)esz;
#pragma unroll
    for (int u = 0; u < 8; ++u) s.v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)(u * step), 2);
  } else {
    const uint32_t step = 8u * a.row_stride_f * (uint32_t)esz;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      s.v[u] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)(u * step), 2));
  }
}
// Sample mode: local row i of sample tile `tile` is index row sample_row(i) = ((u * n_tiles + tile) * kR + i0) * gap with
// i = u * kR + i0 and kR = the rows one load instruction of the producers covers (8 for f32 rows, 16 for bf16): a run
// of kR neighbouring sample rows, and the tile's 128 / kR runs spread over the whole index -- the descriptor of piece u
// carries the run's 64-bit base, the thread's 32-bit offset stays inside the run.  With the first stage of a tile the
// thread also looks up whether its rows' tiles lie beyond the norm cap.
template <bool kBf16> __device__ __forceinline__ size_t sample_row(const FlatFilterArgs &a, uint32_t tile, uint32_t i) {
  constexpr uint32_t kR = kBf16 ? 16 : 8;
  return ((size_t)((i / kR) * a.n_tiles + tile) * kR + i % kR) * a.sample_gap;
}
template <bool kBf16, bool kL2>
__device__ __forceinline__ void ws_rows_load_sample(WsRows<kBf16> &s, const FlatFilterArgs &a, uint32_t tile, uint32_t st, uint32_t t,
                                                    uint32_t cap_bits) {
  constexpr size_t esz = kBf16 ? 2 : 4;
  constexpr int kPieces = kBf16 ? 8 : 16;
  constexpr uint32_t kR = kBf16 ? 16 : 8;
  const size_t row_bytes = (size_t)a.row_stride_f * esz;
  if constexpr (kL2) s.hn = a.hn16[sample_row<kBf16>(a, tile, t)];
  const uint32_t i0 = kBf16 ? t >> 3 : t >> 4;
  const uint32_t voff = i0 * a.sample_gap * (uint32_t)row_bytes + (kBf16 ? (t & 7) : (t & 15)) * 16u;
  const char *base = static_cast<const char *>(a.rows) + (size_t)st * kFStageK * esz;
  // (the tile look-ups are requested BEFORE the rows: loads return in order, so waiting for them leaves the rows in flight)
  uint32_t pz = 0;
  if (st == 0) {
    uint32_t tn[kPieces];
#pragma unroll
    for (int u = 0; u < kPieces; ++u)
      tn[u] = a.tile_norm[((size_t)((uint32_t)u * a.n_tiles + tile) * kR + i0) * a.sample_gap >> 7];
#pragma unroll
    for (int u = 0; u < kPieces; ++u) pz |= (tn[u] > cap_bits ? 1u : 0u) << u;
  }
  s.pz = pz;
#pragma unroll
  for (int u = 0; u < kPieces; ++u) {
    const size_t run = (size_t)((uint32_t)u * a.n_tiles + tile) * kR * a.sample_gap;       // first index row of the run
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc(base + run * row_bytes), (int)voff, 0, 2);
    if constexpr (kBf16) s.v[u] = v;
    else s.v[u] = __builtin_bit_cast(f32x4v, v);
  }
}
// -> f16 in LDS.  f32 rows: round to nearest even.  bf16 rows: bf16 -> f32 is a shift and f32 -> f16 is then EXACT for
// every value in f16's normal range (8 significant bits fit 11), so a bf16 index carries no row rounding error at all.
template <bool kBf16, bool kL2, bool kSample, bool kBfMma = false>
__device__ __forceinline__ void ws_rows_store(_Float16 *buf, uint32_t *hn_buf, uint32_t t, const WsRows<kBf16> &s, bool raw = false) {
  if constexpr (kL2) hn_buf[t] = s.hn;
  if constexpr (kBf16) {
    _Float16 *dst = buf + (t >> 3) * kFAStride + (t & 7) * 8;
    if (kBfMma || raw) {   // bf16 matrix-core path: the bytes as they are (raw: the same as an experiment of the f16 path)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        u32x4v v = s.v[u];
        if constexpr (kSample) {
          if ((t & 7) == 0 && ((s.pz >> u) & 1u)) v[0] = (v[0] & 0xFFFF0000u) | 0x7FC0u;   // bf16 NaN: no witness
        }
        *reinterpret_cast<u32x4v *>(dst + u * 16 * kFAStride) = v;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f16x8 h;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        h[2 * w] = (_Float16)__uint_as_float(s.v[u][w] << 16);
        h[2 * w + 1] = (_Float16)__uint_as_float(s.v[u][w] & 0xFFFF0000u);
      }
      if constexpr (kSample) {
        if ((t & 7) == 0 && ((s.pz >> u) & 1u)) h[0] = __builtin_bit_cast(_Float16, (uint16_t)0x7E00);   // NaN: no witness
      }
      *reinterpret_cast<f16x8 *>(dst + u * 16 * kFAStride) = h;
    }
  } else {
    _Float16 *dst = buf + (t >> 4) * kFAStride + (t & 15) * 4;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f16x4 h;
      h[0] = (_Float16)s.v[u][0];
      h[1] = (_Float16)s.v[u][1];
      h[2] = (_Float16)s.v[u][2];
      h[3] = (_Float16)s.v[u][3];
      if constexpr (kSample) {
        if ((t & 15) == 0 && ((s.pz >> u) & 1u)) h[0] = __builtin_bit_cast(_Float16, (uint16_t)0x7E00);   // NaN: no witness
      }
      *reinterpret_cast<f16x4 *>(dst + u * 8 * kFAStride) = h;
    }
  }
}
// query producer wave p: the B operands of query tiles 4p .. 4p+3 of stage st, 16 x (64 lanes x 16 B), from the
// fragment-major query copy; in LDS a stage is [query tile 8][K-step 4][lane 64] x 16 B
constexpr int kWsBStage = 8 * 4 * kWave;   // in 16-byte slots
constexpr int kWsThreads = 512;
__device__ __forceinline__ void ws_b_load(WsB &b, const FlatFilterArgs &a, uint32_t p, uint32_t st, uint32_t lane) {
  const uint32_t ks_n = a.row_stride_f / 16;
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const uint32_t jt0 = p * 4 + t4, jt = jt0 < a.nqt ? jt0 : a.nqt - 1;
    const __amdgpu_buffer_rsrc_t r = ws_rsrc(reinterpret_cast<const char *>(a.q16) + ((size_t)jt * ks_n + st * 4) * kWave * 16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b.v[t4 * 4 + kk] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(lane * 16 + kk * 1024), 0, 0);
  }
}
__device__ __forceinline__ void ws_b_store(uint4 *slot, uint32_t p, uint32_t lane, const WsB &b) {
  u32x4v *dst = reinterpret_cast<u32x4v *>(slot) + (p * 16) * kWave + lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[i * kWave] = b.v[i];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool kBfMma> __device__ __forceinline__ f32x16 ws_mfma(f16x8 x, f16x8 y, f32x16 c) {
  if constexpr (kBfMma) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
}
// kBfMma (bf16 rows, inner-product space): rows and queries stay bf16 -- no conversion on the way into LDS, the bf16
// matrix-core instruction, a query rounding of 2^-8 in the margin (flat_qprep_kernel)
// kDma (with kBfMma, final pass): the rows go HBM -> LDS directly (buffer_load ... lds), no registers and no LDS store
// instructions in between.  A stage is 128 rows x 128 B without padding; the 16-byte piece c of row r sits at piece
// c ^ ((r >> 1) & 7) of the row (the DMA writes a wave's 64 x 16 B to consecutive LDS bytes, so the swizzle is applied to
// the SOURCE address of each lane; with it the consumers' ds_read_b128 are conflict-free).  A ring of kDmaRing stages: while
// the consumers read stage S, stages S+1 .. S+4 are landing or in flight.
constexpr int kDmaRing = 5;
constexpr uint32_t kDmaStageBytes = kFTileRows * 128;
// kBDma: the B operands go L2 -> LDS by DMA as well -- a (query tile, K-step) block of the fragment-major query copy is
// 1 KB, one DMA piece, and lands as the consumers read it; a ring of three B stages (the stage being read, the next one
// landed or landing, one more in flight: the same two iterations between request and use as through three register sets)
// and no registers, no LDS store instructions (which were most of what the B operands cost a stage: the VGPR -> LDS path
// moves ~80 B per clock, 410 clocks for a 32 KB stage).  The query producers only issue the requests.
constexpr int kBRing = 3;
template <bool kBf16, bool kL2, bool kTiming, bool kSample, int kAbl = 0, bool kBfMma = false, bool kDma = false, bool kBDma = false>
__device__ __forceinline__ void flat_filter_body(const FlatFilterArgs &a) {
  static_assert(!kBDma || (!kDma && !kTiming && !kSample && kAbl == 0), "B by DMA: rows through registers, final pass, no experiment variants");
  static_assert(!kDma || (kBfMma && kBf16 && !kL2 && !kSample && !kTiming), "the DMA row path serves the bf16 final pass");
  // (experiment kernels only, kAbl != 0: a.ablate switches pieces of the pipeline OFF -- results invalid, times tell what bounds a stage:
  //  1 B producers do not store, 2 nor load; 4 row producers do not store, 8 nor load; kAbl 16 / 32: the consumers keep
  //  the B / A fragments they read first)
  const uint32_t abl = kAbl != 0 ? exp_ablate(a) : 0u;
  extern __shared__ _Float16 lds_a[];
  constexpr uint32_t kBufHalfs = kFTileRows * kFAStride;                    // one A stage
  uint4 *lds_b = reinterpret_cast<uint4 *>(lds_a + (kDma ? kDmaRing * kDmaStageBytes / 2 : 2 * kBufHalfs));   // [2][kWsBStage]
  uint32_t *lds_ring = reinterpret_cast<uint32_t *>(lds_b + (kBDma ? kBRing : 2) * kWsBStage); // [4 consumer waves][3][64]
  uint32_t *hn_lds = lds_ring + 4 * kRingWords;                              // [2][128] (kL2)
  // [2]: cancellation seen during tile T -> word T & 1.  (An LDS-qualified pointer: through a generic one the accesses
  // become FLAT instructions, whose out-of-order return forces every later wait for a load to be vmcnt(0).)
  volatile __attribute__((address_space(3))) uint32_t *lds_stop =
      (volatile __attribute__((address_space(3))) uint32_t *)(hn_lds + 2 * kFTileRows);
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t stages = a.row_stride_f / kFStageK;

  // a block's tiles: a contiguous range of the launch's tile sequence (sample mode: of the sample tiles, and what the
  // stream positions below call a row is then the sample tile's number)
  const uint32_t n_tiles = a.n_tiles;
  const uint32_t t_base = n_tiles / gridDim.x, t_rem = n_tiles % gridDim.x;
  uint32_t first_tile = blockIdx.x * t_base + (blockIdx.x < t_rem ? blockIdx.x : t_rem);
  uint32_t my_tiles = t_base + (blockIdx.x < t_rem ? 1u : 0u);
  if constexpr (!kSample) {
    if (a.part_tiles != 0) {   // the early pass: the head of the range; the main pass: what is behind it
      const uint32_t lo = a.part_first < my_tiles ? a.part_first : my_tiles;
      const uint32_t n = a.part_tiles < my_tiles - lo ? a.part_tiles : my_tiles - lo;
      first_tile += lo;
      my_tiles = n;
    }
  }
  if (my_tiles == 0) return;
  const uint32_t row_step = kSample ? 1u : (uint32_t)kFTileRows;
  const uint32_t total = my_tiles * stages;
  uint32_t st_c = 0, tile_c = 0, left_c = total;
  bool stop = false;
  if (tid < 2) lds_stop[tid] = 0;
  if constexpr (kAbl != 0) {
    if (abl != 0) {   // (stages that are never written must not hold NaNs: every pair would pass the gate)
      constexpr uint32_t a_bytes = kDma ? kDmaRing * kDmaStageBytes : 2 * kBufHalfs * 2;
      for (uint32_t i = tid; i < (a_bytes + 2 * kWsBStage * 16) / 16; i += kWsThreads) reinterpret_cast<uint4 *>(lds_a)[i] = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();

  // Leaving early (cancellation) must be decided identically by all eight waves or the next barrier never completes: the
  // polling thread publishes what it saw during tile T in lds_stop[T & 1] before the tile's last barrier, everybody reads
  // that word after it, and the word is not written again before tile T+2, i.e. behind a barrier that follows every read.
  // (The read is hidden from the compiler: it would make every LDS read wait for the LDS-DMA pieces in flight.)
#define VK_WS_TILE_END(BARRIER)                                                                                     \
  {                                                                                                                 \
    BARRIER;                                                                                                        \
    uint32_t sv_;                                                                                                   \
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(sv_) : "v"((uint32_t)(uintptr_t)(lds_stop + (tile_c & 1))) : "memory"); \
    stop = sv_ != 0;                                                                                                \
    st_c = 0;                                                                                                       \
    tile_c += 1;                                                                                                    \
  }

  // (kTiming: cycles per phase, summed over the waves into a.dbg -- row producers 0 loads issued, 2 wait + convert + LDS
  // stores, 3 barrier; query producers 4 issue + wait + LDS stores, 8 barrier; consumers 5 operands + MFMAs, 6 gate,
  // 7 barrier)
  constexpr bool timing = kTiming;
#define VK_WS_TICK(I)                                                                                               \
  if constexpr (timing) {                                                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter();                                                   \
    ph[I] += now_ - tlast;                                                                                          \
    tlast = now_;                                                                                                   \
  }
  // The producers' barrier is the bare instruction behind an explicit wait for the wave's own LDS stores:
  // __syncthreads() puts a release fence in front of it, which for a wave with loads in flight may become vmcnt(0) --
  // every stage of prefetch drained per stage.
#define VK_WS_PBARRIER()                                                                                            \
  {                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_s_barrier();                                                                                   \
    asm volatile("" ::: "memory");                                                                                  \
  }
  // wave priorities (VK_FILTER_PRIO = row producers | query producers << 2 | consumers << 4): the two waves of a SIMD share
  // its issue slots by priority, then age -- and the producers are the younger half of the block
  {
    const uint32_t pr = (exp_prio(a) >> (wave >= 6 ? 2 : wave >= 4 ? 0 : 4)) & 3u;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  if constexpr (kBDma) {
    if (wave >= 6) {
      // ================================ query producer, LDS-DMA ========================================
      // wave p: query tiles 4p .. 4p+3, sixteen 1 KB pieces per stage.  Iteration S: request B(S+2) into the ring slot
      // B(S-1) left, wait for B(S+1), barrier.
      constexpr int kTiles = 4;
      const uint32_t p = wave - 6, ks_n = a.row_stride_f / 16;
      __attribute__((address_space(3))) char *bring = (__attribute__((address_space(3))) char *)lds_b;
      FPos lb{first_tile * row_step, 0, total, row_step};
      uint32_t slot_b = 0;                                                   // ring slot of the next request (bytes)
#define VK_WS_BDMA_ISSUE()                                                                                          \
      {                                                                                                             \
        _Pragma("unroll") for (int t4 = 0; t4 < kTiles; ++t4) {                                                     \
          const uint32_t jt0 = p * 4 + t4, jt = jt0 < a.nqt ? jt0 : a.nqt - 1;                                      \
          const __amdgpu_buffer_rsrc_t r_ = ws_rsrc(reinterpret_cast<const char *>(a.q16) + ((size_t)jt * ks_n + lb.st * 4) * kWave * 16); \
          _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (__attribute__((address_space(3))) void *)(bring + slot_b + ((p * 4u + t4) * 4u + kk) * 1024u), \
                                                     16, (int)(lane * 16), (int)(kk * 1024), 0, 0);                 \
        }                                                                                                           \
        fpos_advance(lb, stages);                                                                                   \
        slot_b = slot_b + kWsBStage * 16u == kBRing * kWsBStage * 16u ? 0u : slot_b + kWsBStage * 16u;              \
      }
      VK_WS_BDMA_ISSUE()
      VK_WS_BDMA_ISSUE()
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      VK_WS_PBARRIER()
      while (left_c != 0 && !stop) {
        const bool live = left_c != 0;
        VK_WS_BDMA_ISSUE()
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        left_c -= live ? 1u : 0u;
        st_c += 1;
        if (live && st_c == stages) VK_WS_TILE_END(VK_WS_PBARRIER()) else VK_WS_PBARRIER();
      }
#undef VK_WS_BDMA_ISSUE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (nothing may land in LDS after the block has gone)
      return;
    }
  }
  if (!kDma && !kBDma && wave >= 6) {
    // ================================ query producer =====================================================
    // Stage s lives in register set s % 3.  Iteration S: request B(S+3) into the set B(S) left (written to LDS one iteration
    // ago), write B(S+1) -- requested two iterations ago -- to slot (S+1) & 1, barrier.  Past the end of the stream the
    // loads re-read its last stage (no branch around a load).  (r02 had two sets, i.e. ONE iteration between request
    // and use: over bf16 rows, where a stage is half the HBM time, the L2 round trip of the B block was the stage.)
    const uint32_t p = wave - 6;
    FPos lb{first_tile * row_step, 0, total, row_step};
    WsB b0, b1, b2;
    ws_b_load(b0, a, p, lb.st, lane);
    fpos_advance(lb, stages);
    ws_b_load(b1, a, p, lb.st, lane);
    fpos_advance(lb, stages);
    ws_b_load(b2, a, p, lb.st, lane);
    fpos_advance(lb, stages);
    ws_b_store(lds_b, p, lane, b0);
    VK_WS_PBARRIER()
    uint32_t bpar = 0;                                                       // S & 1
    unsigned long long ph[2] = {0, 0}, tlast = __builtin_readcyclecounter();
#define VK_WS_BPROD(BLOAD, BSTORE)                                                                                  \
    {                                                                                                               \
      const bool live = left_c != 0;                                                                                \
      VK_WS_TICK(1)                                                                                                 \
      if (!(abl & 2u)) ws_b_load(BLOAD, a, p, lb.st, lane);                                                         \
      fpos_advance(lb, stages);                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      bpar ^= 1;                                                                                                    \
      if (!(abl & 1u)) ws_b_store(lds_b + bpar * kWsBStage, p, lane, BSTORE);                                       \
      if constexpr (timing) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
      VK_WS_TICK(0)                                                                                                 \
      left_c -= live ? 1u : 0u;                                                                                     \
      st_c += 1;                                                                                                    \
      if (live && st_c == stages) VK_WS_TILE_END(VK_WS_PBARRIER()) else VK_WS_PBARRIER();                           \
    }
    while (left_c != 0 && !stop) {
      VK_WS_BPROD(b0, b1)
      if (stop) break;
      VK_WS_BPROD(b1, b2)
      if (stop) break;
      VK_WS_BPROD(b2, b0)
    }
#undef VK_WS_BPROD
    if constexpr (timing) {
      if (lane == 0 && exp_dbg(a)) {
        atomicAdd(&exp_dbg(a)[4], ph[0]);
        atomicAdd(&exp_dbg(a)[8], ph[1]);
      }
    }
    return;
  }
  if constexpr (kDma) {
    if (wave >= 4) {
      // ================================ producers of the DMA kernel ====================================
      // With the rows going HBM -> LDS by DMA there is little left of a row producer, and the B operands were what a
      // stage waited for (ablations: 3.45 ms without them, 4.08 with, 10M x 768): all four producer waves do a quarter
      // of both.  Wave pw: the B operands of query tiles 2 pw, 2 pw + 1 (8 loads and 8 LDS stores per stage, three stages
      // in flight) and pieces 4 pw .. 4 pw + 3 of the row stage (piece j = rows 8 j .. 8 j + 7 of the tile, 1 KB: lane l
      // lands on slot l & 7 of row 8 j + l / 8 and therefore fetches the row's piece (l & 7) ^ swizzle(row)).
      // Loads return in order, so the ORDER of the requests decides what a wait costs: B(S+3) is requested BEFORE the rows
      // of stage S+4, and the wait for B(S+1) and the rows of S+1 then leaves rows(S+2), B(S+2), rows(S+3), B(S+3) and
      // rows(S+4) -- 28 requests -- in flight: nothing is waited for earlier than it is needed.  The B stores are inline
      // assembly: behind a DMA in flight the compiler would put vmcnt(0) in front of every LDS store it knows about.
      const uint32_t pw = wave - 4;
      const uint32_t rpart = ((32u * pw + (lane >> 3)) * a.row_stride_f) * 2u;
      const uint32_t voff_e = rpart + (((lane & 7u) ^ (lane >> 4)) * 16u), voff_o = rpart + (((lane & 7u) ^ (4u + (lane >> 4))) * 16u);
      const uint32_t step = 8u * a.row_stride_f * 2u;
      __attribute__((address_space(3))) char *ring0 = (__attribute__((address_space(3))) char *)lds_a;
      const uint32_t b_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)lds_b) + (pw * 8u * kWave + lane) * 16u;
      const uint32_t ks_n = a.row_stride_f / 16;
      FPos ld{first_tile * row_step, 0, total, row_step}, lb = ld;
      uint32_t slot_i = 0;                                                   // ring slot of the next row request (bytes)
      struct B8 { u32x4v v[8]; } b0, b1, b2;
#define VK_WS_DMA_ROWS()                                                                                            \
      {                                                                                                             \
        const __amdgpu_buffer_rsrc_t r_ =                                                                           \
            ws_rsrc(static_cast<const char *>(a.rows) + ((size_t)ld.row0 * a.row_stride_f + (size_t)ld.st * kFStageK) * 2u); \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                               \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (__attribute__((address_space(3))) void *)(ring0 + slot_i + (pw * 4u + u) * 1024u), 16, \
                                                   (int)((u & 1) ? voff_o : voff_e), (int)(u * step), 0, 2);         \
        fpos_advance(ld, stages);                                                                                   \
        slot_i = slot_i + kDmaStageBytes == kDmaRing * kDmaStageBytes ? 0u : slot_i + kDmaStageBytes;               \
      }
#define VK_WS_DMA_BLOAD(B)                                                                                          \
      {                                                                                                             \
        _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                                          \
          const uint32_t jt0 = pw * 2 + t2, jt = jt0 < a.nqt ? jt0 : a.nqt - 1;                                     \
          const __amdgpu_buffer_rsrc_t r_ = ws_rsrc(reinterpret_cast<const char *>(a.q16) + ((size_t)jt * ks_n + lb.st * 4) * kWave * 16); \
          _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                          \
            B.v[t2 * 4 + kk] = __builtin_amdgcn_raw_buffer_load_b128(r_, (int)(lane * 16 + kk * 1024), 0, 0);       \
        }                                                                                                           \
        fpos_advance(lb, stages);                                                                                   \
      }
#define VK_WS_DMA_BSTORE(B, PAR)                                                                                    \
      {                                                                                                             \
        const uint32_t at_ = b_lds + (PAR) * (kWsBStage * 16u);                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                            \
          asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(at_), "v"(B.v[i_]), "n"(i_ * 1024) : "memory");       \
      }
      VK_WS_DMA_BLOAD(b0)
      VK_WS_DMA_ROWS()
      VK_WS_DMA_BLOAD(b1)
      VK_WS_DMA_ROWS()
      VK_WS_DMA_BLOAD(b2)
      VK_WS_DMA_ROWS()
      VK_WS_DMA_ROWS()
      asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
      VK_WS_DMA_BSTORE(b0, 0u)
      VK_WS_PBARRIER()
      uint32_t bpar = 0;                                                     // S & 1
      // iteration S: B(S+3) -> the set B(S) left, rows(S+4) -> the ring slot stage S-1 left; B(S+1) -> LDS slot (S+1) & 1
#define VK_WS_DMA_PROD(BLOAD, BSTORE)                                                                               \
      {                                                                                                             \
        const bool live = left_c != 0;                                                                              \
        if (live && st_c == 0 && lane == 0 && pw == 0 && a.cancel && (tile_c % kCancelPollTiles) == 0) {            \
          if (__hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) lds_stop[tile_c & 1] = 1; \
        }                                                                                                           \
        const bool b_on = !(abl & 2u), r_on = !(abl & 8u);                                                          \
        if (kAbl == 0 || b_on) VK_WS_DMA_BLOAD(BLOAD)                                                               \
        if (kAbl == 0 || r_on) VK_WS_DMA_ROWS()                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        asm volatile("s_waitcnt vmcnt(28)" ::: "memory");                                                           \
        bpar ^= 1;                                                                                                  \
        if (kAbl == 0 || !(abl & 1u)) VK_WS_DMA_BSTORE(BSTORE, bpar)                                                \
        left_c -= live ? 1u : 0u;                                                                                   \
        st_c += 1;                                                                                                  \
        if (live && st_c == stages) VK_WS_TILE_END(VK_WS_PBARRIER()) else VK_WS_PBARRIER();                         \
      }
      while (left_c != 0 && !stop) {
        VK_WS_DMA_PROD(b0, b1)
        if (stop) break;
        VK_WS_DMA_PROD(b1, b2)
        if (stop) break;
        VK_WS_DMA_PROD(b2, b0)
      }
#undef VK_WS_DMA_PROD
#undef VK_WS_DMA_BSTORE
#undef VK_WS_DMA_BLOAD
#undef VK_WS_DMA_ROWS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (nothing may land in LDS after the block has gone)
      return;
    }
  }
  if (!kDma && wave >= 4) {
    // ================================ row producer =======================================================
    // Rows of stage s live in register set s % N (N = 3, bf16: 6).  Iteration S: request the rows of stage S+N into the
    // set stage S left (converted one iteration ago), convert stage S+1 (the loads of S+2 .. S+N stay outstanding) into
    // LDS buffer (S+1) & 1, barrier.  96 KB of rows per CU are in flight.
    const uint32_t t = tid - 256;
    const uint32_t voff = kBf16 ? ((t >> 3) * a.row_stride_f + (t & 7) * 8) * 2u : ((t >> 4) * a.row_stride_f + (t & 15) * 4) * 4u;
    FPos ld{first_tile * row_step, 0, total, row_step};
    // Register sets of row stages in flight: three for f32 rows (96 KB per CU), SIX for bf16 rows -- a bf16 stage is
    // half the bytes, and with three of them in flight (48 KB per CU) the stream was bound by bytes in flight over
    // latency, not by HBM: 256 CUs x 48 KB / ~3 us = 4 TB/s, which is what the r02 kernel reached over bf16 rows.
    WsRows<kBf16> x0, x1, x2, x3, x4, x5;
    x0.hn = x1.hn = x2.hn = x3.hn = x4.hn = x5.hn = 0;
    x0.pz = x1.pz = x2.pz = x3.pz = x4.pz = x5.pz = 0;
    uint32_t cap_bits = 0;
    if constexpr (kSample) cap_bits = *a.norm_cap;
#define VK_WS_ROWS_LOAD(X)                                                                                          \
    {                                                                                                               \
      if constexpr (kSample) ws_rows_load_sample<kBf16, kL2>(X, a, ld.row0, ld.st, t, cap_bits);                    \
      else ws_rows_load<kBf16, kL2>(X, a, ld.row0, ld.st, t, voff);                                                 \
    }
    VK_WS_ROWS_LOAD(x0)
    fpos_advance(ld, stages);
    VK_WS_ROWS_LOAD(x1)
    fpos_advance(ld, stages);
    VK_WS_ROWS_LOAD(x2)
    fpos_advance(ld, stages);
    if constexpr (kBf16) {
      VK_WS_ROWS_LOAD(x3)
      fpos_advance(ld, stages);
      VK_WS_ROWS_LOAD(x4)
      fpos_advance(ld, stages);
      VK_WS_ROWS_LOAD(x5)
      fpos_advance(ld, stages);
    }
    ws_rows_store<kBf16, kL2, kSample, kBfMma>(lds_a, hn_lds, t, x0);
    VK_WS_PBARRIER()
    uint32_t ppar = 0;                                                       // S & 1
    unsigned long long ph[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define VK_WS_PROD(RLOAD, RSTORE)                                                                                   \
    {                                                                                                               \
      const bool live = left_c != 0;                                                                                \
      VK_WS_TICK(3)                                                                                                 \
      if (live && st_c == 0 && t == 0 && a.cancel && (tile_c % kCancelPollTiles) == 0) {                            \
        if (__hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) lds_stop[tile_c & 1] = 1; \
      }                                                                                                             \
      if (!(abl & 8u)) VK_WS_ROWS_LOAD(RLOAD)                                                                       \
      fpos_advance(ld, stages);                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      VK_WS_TICK(0)                                                                                                 \
      ppar ^= 1;                                                                                                    \
      if (!(abl & 4u)) ws_rows_store<kBf16, kL2, kSample, kBfMma>(lds_a + ppar * kBufHalfs, hn_lds + ppar * kFTileRows, t, RSTORE, (abl & 512u) != 0); \
      else if (abl & 1024u) {   /* (experiment: wait for the stage's rows, neither convert nor store them) */      \
        _Pragma("unroll") for (int u_ = 0; u_ < (kBf16 ? 8 : 16); ++u_) asm volatile("" ::"v"(RSTORE.v[u_]));       \
      }                                                                                                             \
      if constexpr (timing) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
      VK_WS_TICK(2)                                                                                                 \
      left_c -= live ? 1u : 0u;                                                                                     \
      st_c += 1;                                                                                                    \
      if (live && st_c == stages) VK_WS_TILE_END(VK_WS_PBARRIER()) else VK_WS_PBARRIER();                           \
    }
    while (left_c != 0 && !stop) {
      VK_WS_PROD(x0, x1)
      if (stop) break;
      VK_WS_PROD(x1, x2)
      if (stop) break;
      if constexpr (kBf16) {
        VK_WS_PROD(x2, x3)
        if (stop) break;
        VK_WS_PROD(x3, x4)
        if (stop) break;
        VK_WS_PROD(x4, x5)
        if (stop) break;
        VK_WS_PROD(x5, x0)
      } else {
        VK_WS_PROD(x2, x0)
      }
    }
#undef VK_WS_PROD
#undef VK_WS_ROWS_LOAD
    if constexpr (timing) {
      if (lane == 0 && exp_dbg(a))
        for (int i = 0; i < 4; ++i) atomicAdd(&exp_dbg(a)[i], ph[i]);
    }
    return;
  }
#undef VK_WS_PBARRIER

  // ================================== consumer ===========================================================
  const uint32_t li = lane & 31, g = lane >> 5;
  SurvivorRing ring;
  ring.ent = lds_ring + wave * kRingWords;
  ring.cnt = 0;
  const bool has_q = wave * 2 < a.nqt;
  // per query tile of the wave: the column's gate (bound + error polynomial), or in sample mode its margin at the norm cap
  GateCol<kL2> col[2];
  float wit[2];
  bool closed[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const bool have = wave * 2 + t2 < a.nqt;
    const uint32_t jc = have ? (wave * 2 + t2) * 32 + li : 0u;
    const float4 co = a.qcoef[jc];
    closed[t2] = !have || co.w != 0.f;
    if constexpr (kSample) {
      wit[t2] = a.qwit[jc] + 0x1p-21f;
      col[t2].c1 = col[t2].c0 = col[t2].bound = 0.f;
      if constexpr (kL2) col[t2].c2 = 0.f;
    } else {
      wit[t2] = 0.f;
      if constexpr (kL2) col[t2].c2 = co.x;
      col[t2].c1 = co.y;
      col[t2].c0 = co.z;
      col[t2].bound = (closed[t2] || abl != 0) ? __builtin_inff() : a.qbound[jc];   // (experiments: nothing survives)
    }
  }
  f32x16 acc[2][4];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[t2][rt] = zero;
  uint32_t tile_row0 = first_tile * row_step;
  uint32_t par = 0;
  // The tile's norm bound (largest |row|, not squared): ONE vector-memory word per tile and wave, requested when the tile starts and looked at in
  // its gate twelve stages later (a buffer load on purpose: a scalar load shares the LDS reads' counter, and every
  // wait for a fragment behind it would become a wait for everything).
  const __amdgpu_buffer_rsrc_t tile_rsrc = ws_rsrc(a.tile_norm);
  uint32_t norm_bits = 0;
  if constexpr (!kSample) norm_bits = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc, 0, (int)((tile_row0 / (uint32_t)kFTileRows) * 4u), 0);
  __syncthreads();                                            // (the prologue's barrier)

  // iteration S: 32 MFMAs of stage S, operands from LDS (A: buffer S & 1, row li of each row tile; B: slot S & 1,
  // query tiles 2w, 2w+1), fetched one K-step ahead; gate at the end of a tile; barrier.  Not unrolled (the buffer and
  // the slot are offsets in a register): one copy of the gate's code.
#define VK_WS_AB(FA, FB0, FB1, KK)                                                                                  \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                  \
    FA[rt] = (kAbl & 32) ? abl_a[rt]                                                                                \
             : kDma      ? *reinterpret_cast<const f16x8 *>(dma_ab + (dma_off ^ ((KK) * 32u)) + rt * 4096)           \
                         : *reinterpret_cast<const f16x8 *>(ab + rt * 32 * kFAStride + (KK) * 16);                  \
  FB0 = (kAbl & 16) ? abl_b[0] : bb[(KK) * kWave];                                                                  \
  FB1 = (kAbl & 16) ? abl_b[1] : bb[(4 + (KK)) * kWave];
#define VK_WS_MM(FA, FB0, FB1)                                                                                      \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                                \
    acc[0][rt] = ws_mfma<kBfMma>(FA[rt], FB0, acc[0][rt]);                                                          \
    acc[1][rt] = ws_mfma<kBfMma>(FA[rt], FB1, acc[1][rt]);                                                          \
  }
  // (first K-step of a tile: the accumulators are not cleared -- 128 register moves -- but started from the constant 0)
#define VK_WS_MMZ(FA, FB0, FB1)                                                                                     \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                                \
    acc[0][rt] = ws_mfma<kBfMma>(FA[rt], FB0, zero);                                                                \
    acc[1][rt] = ws_mfma<kBfMma>(FA[rt], FB1, zero);                                                                \
  }
#define VK_WS_STAGE(MM0)                                                                                            \
  {                                                                                                                 \
    f16x8 fa[4], fb[4], b0, b1, b2, b3;                                                                             \
    VK_WS_AB(fa, b0, b1, 0)                                                                                         \
    VK_WS_AB(fb, b2, b3, 1)                                                                                         \
    MM0(fa, b0, b1)                                                                                                 \
    VK_WS_AB(fa, b0, b1, 2)                                                                                         \
    VK_WS_MM(fb, b2, b3)                                                                                            \
    VK_WS_AB(fb, b2, b3, 3)                                                                                         \
    VK_WS_MM(fa, b0, b1)                                                                                            \
    VK_WS_MM(fb, b2, b3)                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                                                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                                                             \
  }
  f16x8 abl_a[4], abl_b[2];
  if constexpr (kAbl != 0) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) abl_a[rt] = *reinterpret_cast<const f16x8 *>(lds_a + li * kFAStride + g * 8 + rt * 32 * kFAStride);
    abl_b[0] = reinterpret_cast<const f16x8 *>(lds_b)[lane];
    abl_b[1] = reinterpret_cast<const f16x8 *>(lds_b)[lane + kWave];
  }
  // (kDma: byte offset of the lane's fragment of K-step kk inside a ring slot -- row li, piece (2 kk + g) ^ swizzle)
  // (piece 2 kk + g = 2 kk | g, so K-step kk is the offset of K-step 0 with bits 5-6 flipped by kk)
  const uint32_t dma_off = li * 128u + ((g ^ ((li >> 1) & 7u)) * 16u);
  uint32_t dma_slot = 0, b_slot = 0;                          // ring slots being read (kDma: bytes; kBDma: 16-byte units)
  unsigned long long ph[3] = {0, 0, 0}, tlast = __builtin_readcyclecounter();
  while (left_c != 0 && !stop) {
    VK_WS_TICK(2)
    const _Float16 *ab = lds_a + par * kBufHalfs + li * kFAStride + g * 8;
    const char *dma_ab = reinterpret_cast<const char *>(lds_a) + dma_slot;
    const f16x8 *bb = reinterpret_cast<const f16x8 *>(lds_b + (kBDma ? b_slot : par * kWsBStage) + (wave * 2) * 4 * kWave) + lane;
    if (has_q) {
      if (st_c == 0) VK_WS_STAGE(VK_WS_MMZ) else VK_WS_STAGE(VK_WS_MM)
      if constexpr ((kAbl & 128) != 0) VK_WS_STAGE(VK_WS_MM)   // (experiment: a stage's MFMAs twice per barrier)
      if constexpr (kL2) {
        if (st_c + 1 == stages) {   // one more K-step: (hn_hi, hn_lo, 0 ...) x (-1, -1, 0 ...) = - |x|^2 / 2
          const uint4 nb = make_uint4(g == 0 ? 0xBC00BC00u : 0u, 0u, 0u, 0u);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const uint4 na = make_uint4(g == 0 ? hn_lds[par * kFTileRows + rt * 32 + li] : 0u, 0u, 0u, 0u);
            acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, na), __builtin_bit_cast(f16x8, nb), acc[0][rt], 0, 0, 0);
            acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, na), __builtin_bit_cast(f16x8, nb), acc[1][rt], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (timing) asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    VK_WS_TICK(0)
    left_c -= 1;
    st_c += 1;
    par ^= 1;
    if constexpr (kDma) dma_slot = dma_slot + kDmaStageBytes == kDmaRing * kDmaStageBytes ? 0u : dma_slot + kDmaStageBytes;
    if constexpr (kBDma) b_slot = b_slot + kWsBStage == kBRing * kWsBStage ? 0u : b_slot + kWsBStage;
    if (st_c == stages) {
      if (has_q) {
        if constexpr (kSample) {
          sample_max<4, kBf16>(a, acc[0], wit[0], closed[0], tile_row0, (wave * 2) * 32 + li, g);
          sample_max<4, kBf16>(a, acc[1], wit[1], closed[1], tile_row0, (wave * 2 + 1) * 32 + li, g);
        } else if constexpr ((kAbl & 64) != 0) {   // (experiment: no gate; the accumulators stay alive)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) asm volatile("" ::"v"(acc[0][rt]), "v"(acc[1][rt]));
        } else {
          filter_gate<4, false>(a, acc[0], gate_thr<kL2>(col[0], norm_bits), tile_row0, wave * 2, li, g, ring, lane);
          filter_gate<4, false>(a, acc[1], gate_thr<kL2>(col[1], norm_bits), tile_row0, wave * 2 + 1, li, g, ring, lane);
        }
      }
      tile_row0 += row_step;
      // (past the block's last tile this reads a word behind it: the table is padded, the value is not used)
      if constexpr (!kSample) norm_bits = __builtin_amdgcn_raw_buffer_load_b32(tile_rsrc, 0, (int)((tile_row0 / (uint32_t)kFTileRows) * 4u), 0);
      VK_WS_TICK(1)
      VK_WS_TILE_END(__syncthreads())
    } else {
      __syncthreads();
    }
  }
#undef VK_WS_AB
#undef VK_WS_MM
#undef VK_WS_MMZ
#undef VK_WS_STAGE
#undef VK_WS_TILE_END
#undef VK_WS_TICK
  if constexpr (timing) {
    if (lane == 0 && exp_dbg(a))
      for (int i = 0; i < 3; ++i) atomicAdd(&exp_dbg(a)[5 + i], ph[i]);
  }
  ring_flush(a, ring, lane);
}


template <bool kBf16, bool kL2, bool kTiming, int kAbl = 0>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, kTiming, false, kAbl>(a);
}
// bf16 rows on the bf16 matrix cores (inner-product space): final pass and sample pass
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_bfmma_kernel(FlatFilterArgs a) {
  flat_filter_body<true, false, false, false, 0, true>(a);
}
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_bfmma_dma_kernel(FlatFilterArgs a) {
  flat_filter_body<true, false, false, false, 0, true, true>(a);
}
// B operands by DMA (rows through registers): the final pass of every row format and space except the bf16 DMA kernel
template <bool kBf16, bool kL2, bool kBfMma>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_bdma_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, false, false, 0, kBfMma, false, true>(a);
}
// the early pass of a batch that walks the index in two launches (FlatFilterArgs::part_tiles): the final pass's code under its
// own names
template <bool kBf16, bool kL2, bool kBfMma>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_early_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, false, false, 0, kBfMma, false, true>(a);
}
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_early_bfmma_dma_kernel(FlatFilterArgs a) {
  flat_filter_body<true, false, false, false, 0, true, true>(a);
}
#ifdef VK_EXPERIMENTS
template <int kAbl>   // (experiments, VK_FILTER_ABLATE)
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_bfmma_dma_abl_kernel(FlatFilterArgs a) {
  flat_filter_body<true, false, false, false, kAbl, true, true>(a);
}
#endif
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_bfmma_sample_kernel(FlatFilterArgs a) {
  flat_filter_body<true, false, false, true, 0, true>(a);
}
// the pass over the bound's sample (every s-th tile of the index, a few per cent of the rows): the same pipeline, group
// bounds instead of survivors.  (B by DMA here too was tried in r04: same answers, same 0.20 ms at 2048 sample tiles -- a
// sample tile costs 22 us of a CU against 16.7 in the final pass with or without it; not kept)
template <bool kBf16, bool kL2>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_sample_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, false, true>(a);
}

size_t flat_filter_dma_lds_bytes() {
  return (size_t)kDmaRing * kDmaStageBytes + (size_t)2 * kWsBStage * 16 + (size_t)4 * kRingWords * 4 + (size_t)2 * kFTileRows * 4 + 16;
}
size_t flat_filter_bdma_lds_bytes() {
  return (size_t)2 * kFTileRows * kFAStride * sizeof(_Float16) + (size_t)kBRing * kWsBStage * 16 + (size_t)4 * kRingWords * 4 +
         (size_t)2 * kFTileRows * 4 + 16;
}
size_t flat_filter_lds_bytes() {
  return (size_t)2 * kFTileRows * kFAStride * sizeof(_Float16) + (size_t)2 * kWsBStage * 16 + (size_t)4 * kRingWords * 4 +
         (size_t)2 * kFTileRows * 4 + 16;   // (+ the stop words)
}

bool flat_filter_supported(uint32_t row_stride_f, uint64_t k, bool bf16, bool l2) {
  (void)bf16;
  (void)l2;
  return (row_stride_f % kFStageK) == 0 && k >= 1 && k <= 1024;   // (the re-rank has 1, 4 and 16 result slots per lane)
}

hipError_t launch_flat_qprep(const FlatFilterArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(flat_qprep_kernel, dim3(a.nqt * 32), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_flat_filter(const FlatFilterArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.nqt == 0 || a.nqt > 8 || blocks == 0 || a.n_tiles == 0 || (a.mode == 1 && a.sample_gap == 0)) return hipErrorInvalidValue;
  size_t lds = flat_filter_lds_bytes();
  const void *fn = a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_kernel<true, true, false>)
                                  : reinterpret_cast<const void *>(&flat_filter_kernel<true, false, false>))
                          : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_kernel<false, true, false>)
                                  : reinterpret_cast<const void *>(&flat_filter_kernel<false, false, false>));
  if (a.mode == 1)
    fn = a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_sample_kernel<true, true>)
                        : reinterpret_cast<const void *>(&flat_filter_sample_kernel<true, false>))
                : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_sample_kernel<false, true>)
                        : reinterpret_cast<const void *>(&flat_filter_sample_kernel<false, false>));
  if (a.qbf16) {
    if (!a.bf16 || a.l2) return hipErrorInvalidValue;
    fn = a.mode == 1 ? reinterpret_cast<const void *>(&flat_filter_bfmma_sample_kernel) : reinterpret_cast<const void *>(&flat_filter_bfmma_kernel);
    if (a.mode == 0 && a.dma) {
      fn = a.early ? reinterpret_cast<const void *>(&flat_filter_early_bfmma_dma_kernel) : reinterpret_cast<const void *>(&flat_filter_bfmma_dma_kernel);
#ifdef VK_EXPERIMENTS
      if (a.ablate_on) fn = (a.ablate & 112u) == 112u ? reinterpret_cast<const void *>(&flat_filter_bfmma_dma_abl_kernel<113>)
                                                        : reinterpret_cast<const void *>(&flat_filter_bfmma_dma_abl_kernel<1>);
#endif
      lds = flat_filter_dma_lds_bytes();
    }
  }
#ifdef VK_EXPERIMENTS
  const bool exp_variant = a.timing || a.ablate_on;
#else
  constexpr bool exp_variant = false;
#endif
  if (a.bdma && a.mode == 0 && !exp_variant && !(a.qbf16 && a.dma)) {
    fn = a.qbf16 ? reinterpret_cast<const void *>(&flat_filter_bdma_kernel<true, false, true>)
         : a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_bdma_kernel<true, true, false>)
                          : reinterpret_cast<const void *>(&flat_filter_bdma_kernel<true, false, false>))
                  : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_bdma_kernel<false, true, false>)
                          : reinterpret_cast<const void *>(&flat_filter_bdma_kernel<false, false, false>));
    if (a.early)
      fn = a.qbf16 ? reinterpret_cast<const void *>(&flat_filter_early_kernel<true, false, true>)
           : a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_early_kernel<true, true, false>)
                            : reinterpret_cast<const void *>(&flat_filter_early_kernel<true, false, false>))
                    : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_early_kernel<false, true, false>)
                            : reinterpret_cast<const void *>(&flat_filter_early_kernel<false, false, false>));
    lds = flat_filter_bdma_lds_bytes();
  }
#ifdef VK_EXPERIMENTS
  if (a.timing) {
    if (a.l2 || a.mode == 1 || a.qbf16) return hipErrorInvalidValue;
    fn = a.bf16 ? reinterpret_cast<const void *>(&flat_filter_kernel<true, false, true>)
                : reinterpret_cast<const void *>(&flat_filter_kernel<false, false, true>);
  }
  if (a.ablate_on && !(a.qbf16 && a.dma && a.mode == 0)) {   // VK_FILTER_ABLATE: the experiment kernels (results invalid)
    if (a.l2 || a.mode == 1 || a.qbf16) return hipErrorInvalidValue;
#define VK_ABL(T, N) (a.bf16 ? reinterpret_cast<const void *>(&flat_filter_kernel<true, false, T, N>) : reinterpret_cast<const void *>(&flat_filter_kernel<false, false, T, N>))
    switch (a.ablate & 240u) {
      case 240: fn = VK_ABL(false, 241); break;
      case 16: fn = VK_ABL(false, 17); break;
      case 32: fn = VK_ABL(false, 33); break;
      case 48: fn = VK_ABL(false, 49); break;
      case 64: fn = VK_ABL(false, 65); break;
      case 112: fn = VK_ABL(false, 113); break;
      default: fn = a.timing ? VK_ABL(true, 1) : VK_ABL(false, 1); break;
    }
#undef VK_ABL
  }
#endif
  hipError_t e = ensure_max_lds(fn);   // (above the 64 KB default)
  if (e != hipSuccess) return e;
  FlatFilterArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(fn, dim3(blocks), dim3(kWsThreads), params, lds, s);
}

}  // namespace vk
