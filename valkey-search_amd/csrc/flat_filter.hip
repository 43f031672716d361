// flat_filter.hip -- K4h: the candidate stage of the batched FLAT search on the f16 / bf16 matrix cores.
//
// The exact batched kernel (flat_gemm.hip) reproduces the reference's f32 arithmetic on the f32 MFMA pipe, which is
// 1/16 of the f16 rate: 36.7 ms per 256-query batch at 10M x 768, where ONE pass over the rows costs 4.9 ms of HBM
// time.  An exact answer does not need exact arithmetic for every row: this kernel computes an APPROXIMATE dot product
// of every (row, query) pair with v_mfma_f32_32x32x16_f16 (rows converted f32 -> f16 on the way into LDS, queries
// converted once) and keeps only the pairs that can still be among the query's k best:
//
//   approx >= 1 - bound_q - eps_q          (inner-product space: distance = 1 - dot)
//
// where bound_q is a valid upper bound of the query's final k-th best EXACT distance (the exact kernel over a sample of
// the rows provides it) and eps_q bounds |approx - exact| rigorously (below).  The survivors -- a few hundred to a few
// thousand per query out of 10M -- are re-ranked by the exact quad kernel (flat_scan.hip with a row list) and selected
// by (distance,label), so the answer is BIT-IDENTICAL to the exact path's; only the work is different.  If a query's
// survivor list overflows (duplicates of one vector by the hundred thousand, a filter that leaves no bound, values
// outside the f16 range) the launch raises a flag and the exact kernel, enqueued behind it, runs instead -- decided on
// the device, no host round trip.
//
// Error bound.  x^ = rne_f16(x), q^ = rne_f16(q): |x^_i - x_i| <= 2^-11 |x_i| + 2^-25 (the second term covers f16
// subnormals), same for q.  Products of f16 values are exact in f32; the MFMA accumulates in f32, allowed here 4 ulp
// per accumulated term (D * 2^-22 relative to sum |x_i q_i|), far more than an IEEE chain needs.  With
// sum |x_i q_i| <= |x| |q| (Cauchy-Schwarz; tight exactly for the near neighbours that matter):
//   |approx - dot_real| <= |x||q| (2^-11 + 2^-11 + 2^-22 + D 2^-22) + 2^-25 sqrt(D) (|x| + |q|) (1 + 2^-11)
// and the reference's own f32 result differs from dot_real by at most (D/16 + 5) 2^-24 |x||q|, its 1 - dot by 2^-24
// max(1, |dist|).  eps_q below adds these with |x| <= R = the largest row norm in the index (tracked by row_stats_kernel)
// and rounds everything up.
//
// Data movement per 128-row tile and block (4 waves, one per SIMD, 128 rows x 256 queries of accumulators):
//   rows     393 KB f32 from HBM, once, by all 256 threads -> converted -> f16 in LDS (36 KB, two stages) -> A operands
//            of all four waves (ds_read_b128, conflict-free 144-B row stride)
//   queries  wave w owns queries [64w, 64w+64): its B operands come straight from L2 in MFMA fragment order
//            (16 B per lane and K-step, prepared once by flat_qprep_kernel) -- no LDS, no sharing needed
// so HBM traffic is the row bytes and L2 traffic twice that.  Roofline: HBM (30.72 GB per launch at 10M x 768 f32);
// the matrix cores are about one third busy (3.9 TFLOP of f16 per launch against 2.5 PFLOP/s).
#include <stdlib.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace vk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kFTileRows = 128;
constexpr int kFStageK = 64;                 // k per pipeline stage: 4 MFMA K-steps of 16
constexpr int kFAStride = 72;                // halfs per staged row: 64 + 8 pad = 144 B (conflict-free b128 reads)
constexpr uint32_t kF16Safe = 0x47000000u;   // 32768.0f: inputs beyond it do not go through f16
}  // namespace

// ---- row statistics: the largest row norm and the largest |element| over rows [lo, hi) -------------------------------
// stats[0] = max over rows of |x|^2 (f32 bits, rounded up), stats[1] = max |x_i| (f32 bits; +inf for a non-finite element):
// both only ever grow (atomicMax on the bit patterns of non-negative floats), which keeps them valid bounds when rows
// are overwritten or removed.
__global__ __launch_bounds__(256) void row_stats_kernel(const void *rows, uint32_t bf16, uint32_t stride_e, uint32_t chunks,
                                                        uint32_t lo, uint32_t hi, uint32_t *stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 3, rq = lane >> 2;
  const uint32_t total_waves = gridDim.x * 4, n_tiles = (hi - lo + kRowsPerWave - 1) / kRowsPerWave;
  float best_n2 = 0.f, best_abs = 0.f;
  for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += total_waves) {
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
    best_n2 = fmaxf(best_n2, n2);
    best_abs = fmaxf(best_abs, mx);
  }
  best_n2 = wave_max_f32(best_n2) * 1.0001f;   // (rounding of the sum itself: far below 1e-4)
  best_abs = wave_max_f32(best_abs);
  if (lane == 0) {
    atomicMax(&stats[0], __float_as_uint(best_n2));
    atomicMax(&stats[1], __float_as_uint(best_abs));
  }
}

hipError_t launch_row_stats(const void *rows, bool bf16, uint32_t stride_e, uint32_t lo, uint32_t hi, uint32_t *stats, hipStream_t s) {
  if (hi <= lo) return hipSuccess;
  const uint32_t tiles = (hi - lo + kRowsPerWave - 1) / kRowsPerWave;
  const uint32_t blocks = std::min<uint32_t>((tiles + 3) / 4, 2048);
  hipLaunchKernelGGL(row_stats_kernel, dim3(blocks), dim3(256), 0, s, rows, bf16 ? 1u : 0u, stride_e, stride_e / 16, lo, hi, stats);
  return hipGetLastError();
}

// ---- query preparation -------------------------------------------------------------------------------------------------
// One wave per query column of the (padded) batch: f16 copy in MFMA fragment order, and the gate in dot space.
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
    f16x8 h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      h[t] = (_Float16)e[t];                            // round to nearest even
      n2 = fmaf(e[t], e[t], n2);
      mx = fmaxf(mx, fabsf(e[t]));
      bad = bad || !(e[t] - e[t] == 0.f);
    }
    const uint32_t ks = k8 >> 1, g = k8 & 1;
    reinterpret_cast<f16x8 *>(a.q16)[((size_t)(jt * ks_n + ks) * kWave + g * 32 + jj)] = h;
  }
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) {
    n2 += __shfl_xor(n2, m);
    mx = fmaxf(mx, __shfl_xor(mx, m));
    bad = bad || __shfl_xor((int)bad, m);
  }
  if (lane != 0) return;
  float thr = __builtin_inff();                        // padding column: nothing passes
  if (j < a.nq) {
    const float R = sqrtf(__uint_as_float(a.row_stats[0])) * 1.0001f, amax = __uint_as_float(a.row_stats[1]);
    const float qn = sqrtf(n2) * 1.0001f, bound = a.bound[q];
    const float D = (float)a.row_stride_f;
    // see the header: relative part, absolute (subnormal) part, the reference's own rounding, 1 - dot
    const float rel = 0x1p-10f + 0x1p-22f + D * 0x1p-22f + (D / 16.f + 5.f) * 0x1p-24f;
    const float eps = (R * qn * rel + 0x1.01p-25f * sqrtf(D) * (R + qn) + 0x1p-23f * fmaxf(1.f, 1.f + R * qn)) * 1.001f;
    const bool f16_ok = !bad && mx <= 32768.f && amax <= 32768.f && (R - R == 0.f) && (eps - eps == 0.f);
    // no bound yet (fewer than k allowed rows in the sample) or inputs that cannot go through f16: every row passes,
    // the list overflows, and the exact kernel answers this batch
    thr = (f16_ok && bound - bound == 0.f) ? ((1.f - bound) - eps) - 0x1p-22f * fmaxf(1.f, fabsf(1.f - bound)) : -__builtin_inff();
  }
  a.thr[j] = thr;
}

// ---- the filter ------------------------------------------------------------------------------------------------------
struct RowStage { float4 v[8]; };   // this thread's share of one stage: 128 rows x 64 k f32 = 2048 float4 / 256 threads

template <bool kBf16>
__device__ __forceinline__ RowStage stage_rows_load(const FlatFilterArgs &a, uint32_t tile_row0, uint32_t st, uint32_t tid) {
  RowStage s;
  // idx = tid + 256 u: row = idx / 16, 16-byte column idx % 16 of the row's 256-byte stage slice (coalesced 256 B per row)
  const float *base = static_cast<const float *>(a.rows) + (size_t)tile_row0 * a.row_stride_f + st * kFStageK;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t idx = tid + 256u * u;
    s.v[u] = *reinterpret_cast<const float4 *>(base + (size_t)(idx >> 4) * a.row_stride_f + (idx & 15) * 4);
  }
  return s;
}

__device__ __forceinline__ void stage_rows_store(_Float16 *buf, uint32_t tid, const RowStage &s) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t idx = tid + 256u * u;
    f16x4 h;
    h[0] = (_Float16)s.v[u].x;
    h[1] = (_Float16)s.v[u].y;
    h[2] = (_Float16)s.v[u].z;
    h[3] = (_Float16)s.v[u].w;
    *reinterpret_cast<f16x4 *>(buf + (idx >> 4) * kFAStride + (idx & 15) * 4) = h;
  }
}

struct BFrags { f16x8 b[2][4]; };   // this wave's two query tiles x the stage's four K-steps

__device__ __forceinline__ BFrags stage_b_load(const FlatFilterArgs &a, uint32_t wave, uint32_t st, uint32_t lane) {
  BFrags f;
  const uint32_t ks_n = a.row_stride_f / 16;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint32_t jt = wave * 2 + t < a.nqt ? wave * 2 + t : a.nqt - 1;   // (a wave without queries re-reads the last tile)
    const f16x8 *p = reinterpret_cast<const f16x8 *>(a.q16) + ((size_t)(jt * ks_n + st * 4) * kWave + lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.b[t][kk] = p[(size_t)kk * kWave];
  }
  return f;
}

// Tile done: the gate.  Output register r of row tile rt is row rt*32 + (r&3) + 8*(r>>2) + 4*g, column li of the wave's
// query tile t.  Almost every 32 x 32 block has no survivor: one max over the lane's 16 values, one ballot.
__device__ __forceinline__ void filter_gate(const FlatFilterArgs &a, f32x16 (&acc)[4][2], const float (&thr)[2], uint32_t tile_row0,
                                            uint32_t wave, uint32_t li, uint32_t g) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      float m = acc[rt][t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rt][t][r]);
      if (__builtin_amdgcn_ballot_w64(m >= thr[t]) != 0) {
        const uint32_t q = (wave * 2 + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = tile_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (acc[rt][t][r] >= thr[t] && row < a.n_rows && q < a.nq) {
            if (a.allow_bits == nullptr || allow_bit(a.allow_bits, a.allow_nbits, a.labels[row])) {
              const uint32_t at = atomicAdd(&a.cand_cnt[q], 1u);
              if (at < a.cap) a.cand_row[(size_t)q * a.cap + at] = row;
              else __hip_atomic_store(a.ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
      acc[rt][t] = zero;
    }
  }
}

__global__ __launch_bounds__(256, 1) void flat_filter_kernel(FlatFilterArgs a) {
  extern __shared__ _Float16 lds_a[];   // [2 stages][128 rows][kFAStride]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t li = lane & 31, g = lane >> 5;
  const uint32_t stages = a.row_stride_f / kFStageK;
  constexpr uint32_t kBufHalfs = kFTileRows * kFAStride;

  // this block's contiguous range of row tiles
  const uint32_t n_tiles = (a.n_rows + kFTileRows - 1) / kFTileRows;
  const uint32_t t_base = n_tiles / gridDim.x, t_rem = n_tiles % gridDim.x;
  const uint32_t first_tile = blockIdx.x * t_base + (blockIdx.x < t_rem ? blockIdx.x : t_rem);
  const uint32_t my_tiles = t_base + (blockIdx.x < t_rem ? 1u : 0u);
  if (my_tiles == 0) return;
  const uint32_t total = my_tiles * stages;
  const bool has_q = wave * 2 < a.nqt;

  // gates of this lane's two query columns
  float thr[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) thr[t] = wave * 2 + t < a.nqt ? a.thr[(wave * 2 + t) * 32 + li] : __builtin_inff();

  // flattened (tile, stage) stream; loads run two stages ahead, LDS one stage ahead
  auto pos_of = [&](uint32_t s, uint32_t &row0, uint32_t &st) {
    const uint32_t sc = s < total ? s : total - 1;          // past the end: re-read the last stage (unused)
    row0 = (first_tile + sc / stages) * kFTileRows;
    st = sc % stages;
  };
  uint32_t r0, s0;
  pos_of(0, r0, s0);
  // register sets by stage parity: ra holds even stages, rb odd ones; b0 / b1 the same for the B operands.  The loop is
  // unrolled by two with the sets named explicitly -- rotating them through a copy would make the copy wait for loads
  // that are still in flight.
  RowStage ra = stage_rows_load<false>(a, r0, s0, tid), rb;
  stage_rows_store(lds_a, tid, ra);
  BFrags b0 = stage_b_load(a, wave, s0, lane), b1 = b0;
  pos_of(1, r0, s0);
  rb = stage_rows_load<false>(a, r0, s0, tid);
  __syncthreads();

  f32x16 acc[4][2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) { acc[rt][0] = zero; acc[rt][1] = zero; }

  uint32_t tile_row0 = first_tile * kFTileRows;
  uint32_t cancel_now = 0;
  bool stop = false;

  // iteration S: HBM loads of stage S+2 into RLOAD (its previous content, stage S, went to LDS one iteration ago), B
  // operands of stage S+1 into BNEXT, the 32 MFMAs of stage S (A from LDS buffer S & 1, B from BCUR), stage S+1 from
  // RSTORE into the other LDS buffer, one barrier.  (S == total only when the stream has an odd length: no compute.)
#define VK_FSTAGE(S, RLOAD, RSTORE, BCUR, BNEXT)                                                                    \
  {                                                                                                                 \
    const uint32_t s_ = (S);                                                                                        \
    const uint32_t st = s_ % stages;                                                                                \
    const bool live = s_ < total;                                                                                   \
    if (live && st == 0 && a.cancel && ((s_ / stages) % kCancelPollTiles) == 0)                                     \
      cancel_now = __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                        \
    uint32_t nr0, nst, n1r0, n1st;                                                                                  \
    pos_of(s_ + 2, nr0, nst);                                                                                       \
    pos_of(s_ + 1, n1r0, n1st);                                                                                     \
    RLOAD = stage_rows_load<false>(a, nr0, nst, tid);                                                               \
    BNEXT = stage_b_load(a, wave, n1st, lane);                                                                      \
    const _Float16 *ab = lds_a + (s_ & 1) * kBufHalfs + li * kFAStride + g * 8;                                     \
    if (has_q && live) {                                                                                            \
      _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                            \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                          \
          const f16x8 av = *reinterpret_cast<const f16x8 *>(ab + rt * 32 * kFAStride + kk * 16);                    \
          acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, BCUR.b[0][kk], acc[rt][0], 0, 0, 0);              \
          acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, BCUR.b[1][kk], acc[rt][1], 0, 0, 0);              \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    stage_rows_store(lds_a + ((s_ + 1) & 1) * kBufHalfs, tid, RSTORE);                                              \
    if (live && st + 1 == stages) {                                                                                 \
      if (has_q) filter_gate(a, acc, thr, tile_row0, wave, li, g);                                                  \
      tile_row0 += kFTileRows;                                                                                      \
      stop = __syncthreads_or((int)cancel_now) != 0;   /* block-uniform: every wave leaves at the same tile */      \
    } else {                                                                                                        \
      __syncthreads();                                                                                              \
    }                                                                                                               \
  }

  for (uint32_t s = 0; s < total && !stop; s += 2) {
    VK_FSTAGE(s, ra, rb, b0, b1)
    if (stop) break;
    VK_FSTAGE(s + 1, rb, ra, b1, b0)
  }
#undef VK_FSTAGE
}

size_t flat_filter_lds_bytes() { return (size_t)2 * kFTileRows * kFAStride * sizeof(_Float16); }

bool flat_filter_supported(uint32_t row_stride_f, uint64_t k, bool bf16, bool l2) {
  return !bf16 && !l2 && (row_stride_f % kFStageK) == 0 && k >= 1 && k <= 64;
}

hipError_t launch_flat_qprep(const FlatFilterArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(flat_qprep_kernel, dim3(a.nqt * 32), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_flat_filter(const FlatFilterArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.nqt == 0 || a.nqt > 8 || blocks == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(flat_filter_kernel, dim3(blocks), dim3(256), flat_filter_lds_bytes(), s, a);
  return hipGetLastError();
}

}  // namespace vk
