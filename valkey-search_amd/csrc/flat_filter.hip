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
// Data movement per 128-row tile and block (eight waves with three different jobs, see the kernel):
//   rows     393 KB f32 from HBM, once -> converted -> f16 in LDS (two stages of 18 KB, 144-B row stride: conflict-free
//            ds_read_b128) -> A operands of the four multiplying waves
//   queries  the batch's f16 copy in MFMA fragment order (flat_qprep_kernel, 384 KB at 256 x 768) comes from L2, one
//            32 KB stage at a time, through LDS -> B operands
// so HBM traffic is the row bytes and L2 traffic twice that (PMC: TCC misses 30.8 GB, hits 30.8 GB per launch).
// Roofline: HBM (30.72 GB per launch at 10M x 768 f32); the matrix cores are about one third busy (3.9 TFLOP of f16
// per launch against 2.5 PFLOP/s).
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
}  // namespace

// ---- row statistics: the largest row norm and the largest |element| over rows [lo, hi) -------------------------------
// stats[0] = max over rows of |x|^2 (f32 bits, rounded up), stats[1] = max |x_i| (f32 bits; +inf for a non-finite element):
// both only ever grow (atomicMax on the bit patterns of non-negative floats), which keeps them valid bounds when rows
// are overwritten or removed.
// hn16 (optional, L2 indexes): per row half its squared norm, split into two f16 (hi | lo << 16) -- the extra K-step
// that turns the filter's dot product into dot - |x|^2 / 2.
__global__ __launch_bounds__(256) void row_stats_kernel(const void *rows, uint32_t bf16, uint32_t stride_e, uint32_t chunks,
                                                        uint32_t lo, uint32_t hi, uint32_t *stats, uint32_t *hn16) {
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
    if (hn16 != nullptr && row < hi && j == 0) {
      const float hn = 0.5f * n2;
      const _Float16 h0 = (_Float16)hn;
      const _Float16 h1 = (_Float16)(hn - (float)h0);
      hn16[row] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    }
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

hipError_t launch_row_stats(const void *rows, bool bf16, uint32_t stride_e, uint32_t lo, uint32_t hi, uint32_t *stats, uint32_t *hn16,
                            hipStream_t s) {
  if (hi <= lo) return hipSuccess;
  const uint32_t tiles = (hi - lo + kRowsPerWave - 1) / kRowsPerWave;
  const uint32_t blocks = std::min<uint32_t>((tiles + 3) / 4, 2048);
  hipLaunchKernelGGL(row_stats_kernel, dim3(blocks), dim3(256), 0, s, rows, bf16 ? 1u : 0u, stride_e, stride_e / 16, lo, hi, stats, hn16);
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
    const float rel = (a.bf16 ? 0x1p-11f : 0x1p-10f) + 0x1p-22f + D * 0x1p-22f + (D / 16.f + 5.f) * 0x1p-24f;   // (bf16 rows convert exactly)
    const float eps = (R * qn * rel + 0x1.01p-25f * sqrtf(D) * (R + qn) + 0x1p-23f * fmaxf(1.f, 1.f + R * qn)) * 1.001f;
    bool f16_ok = !bad && mx <= 32768.f && amax <= 32768.f && (R - R == 0.f) && (eps - eps == 0.f);
    // no bound yet (fewer than k allowed rows in the sample: the filter leaves only a few rows of the whole index): the
    // gate is open, every ALLOWED row becomes a survivor and the re-rank settles it
    if (!a.l2) {
      thr = (bound - bound == 0.f) ? ((1.f - bound) - eps) - 0x1p-22f * fmaxf(1.f, fabsf(1.f - bound)) : -__builtin_inff();
    } else {
      // |x - q|^2 = 2 (|x|^2/2) + |q|^2 - 2 x.q: the kernel accumulates x.q - |x|^2/2 (half norms as one more K-step,
      // split in two f16: 2^-21 relative), so a row stays iff  acc >= (|q|^2 - bound - eps2) / 2.  eps2: twice the dot
      // product's margin, the f32 rounding of both norms (D 2^-23 relative, generously), the split of the half norm, and
      // the reference's own rounding of its sum of squared differences ((D/16 + 6) 2^-24 of at most (R + |q|)^2).
      const float nq2 = n2;
      const float sumsq = R * R + qn * qn, top = (R + qn) * (R + qn);
      const float eps2 = (2.f * eps + sumsq * (D * 0x1p-23f + 0x1p-20f) + top * (D / 16.f + 6.f) * 0x1p-23f) * 1.001f;
      f16_ok = f16_ok && 0.5f * R * R <= 60000.f && (eps2 - eps2 == 0.f);
      const float c = 0.5f * (nq2 - bound) - 0.5f * eps2;
      thr = (bound - bound == 0.f) ? c - 0x1p-21f * fmaxf(1.f, fabsf(c)) : -__builtin_inff();
    }
    // Inputs that cannot go through f16 (values beyond its range, non-finite rows or queries, half norms beyond 60000):
    // the products may be NaN, and a NaN passes no gate, open or not -- so the hand-over to the exact kernel is
    // requested here, outright, and this column's gate stays closed.
    if (!f16_ok) {
      thr = __builtin_inff();
      __hip_atomic_store(a.ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  a.thr[j] = thr;
}

// ---- survivors, gate, stream position --------------------------------------------------------------------------------
// Survivors are collected per wave in LDS (a ring of 64 (query, row) entries) and written out 64 at a time: the global
// append is an atomicAdd that RETURNS the slot, a round trip of a microsecond or two -- paid per survivor it sat on the
// critical path of every row tile (some wave of the block nearly always had one, and the block's barrier waits for it).
struct SurvivorRing {
  uint32_t *q;     // [64] LDS
  uint32_t *row;   // [64] LDS
  uint32_t cnt;    // wave-uniform
};
__device__ __forceinline__ void ring_flush(const FlatFilterArgs &a, SurvivorRing &r, uint32_t lane) {
  if (lane < r.cnt) {
    const uint32_t q = r.q[lane], row = r.row[lane];
    const uint32_t at = atomicAdd(&a.cand_cnt[q], 1u);
    if (at < a.cap) a.cand_row[(size_t)q * a.cap + at] = row;
    else __hip_atomic_store(a.ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  r.cnt = 0;
}

// Tile done: the gate.  Output register r of row tile rt is row rt*32 + (r&3) + 8*(r>>2) + 4*g, column li of the wave's
// query tile.  Almost every 32 x 32 block has no survivor: one max over the lane's 16 values, one ballot.
template <int kRt, bool kZero>
__device__ __forceinline__ void filter_gate(const FlatFilterArgs &a, f32x16 (&acc)[kRt], float thr, uint32_t tile_row0,
                                            uint32_t wave, uint32_t li, uint32_t g, SurvivorRing &ring, uint32_t lane) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < kRt; ++rt) {
    float m = acc[rt][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rt][r]);
    if (__builtin_amdgcn_ballot_w64(m >= thr) != 0) {   // (rare: a survivor somewhere in this 32 x 32 block)
      const uint32_t q = wave * 32 + li;
      // the lane's passing registers as a bit mask, then one round per remaining bit of the busiest lane (usually one)
      uint32_t mk = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) mk |= acc[rt][r] >= thr ? 1u << r : 0u;
      if (q >= a.nq) mk = 0;
      while (__builtin_amdgcn_ballot_w64(mk != 0) != 0) {
        const uint32_t r = (uint32_t)__builtin_ctz(mk | 0x10000u);
        const uint32_t row = tile_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        bool pass = mk != 0 && row < a.n_rows;
        if (pass && a.allow_bits != nullptr) pass = allow_bit(a.allow_bits, a.allow_nbits, a.labels[row]);
        mk &= mk - 1;
        const uint64_t pm = __builtin_amdgcn_ballot_w64(pass);
        if (pm != 0) {
          const uint32_t n = (uint32_t)__popcll(pm);
          if (ring.cnt + n > kWave) ring_flush(a, ring, lane);
          if (pass) {
            const uint32_t at = ring.cnt + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
            ring.q[at] = q;
            ring.row[at] = row;
          }
          ring.cnt += n;
        }
      }
    }
    if constexpr (kZero) acc[rt] = zero;
  }
}

// position in a block's flattened (tile, stage) stream, advanced without divisions; it never moves past the last
// stage (prefetches behind the end re-read it and are not used)
struct FPos { uint32_t row0, st, left; };
__device__ __forceinline__ void fpos_advance(FPos &p, uint32_t stages) {
  const bool go = p.left > 1;
  const bool wrap = go && p.st + 1 == stages;
  p.left -= p.left != 0 ? 1u : 0u;
  p.st = wrap ? 0u : p.st + (go ? 1u : 0u);
  p.row0 += wrap ? (uint32_t)kFTileRows : 0u;
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
template <bool kBf16> struct WsRows { f32x4v v[16]; uint32_t hn; };    // row producer thread: 128 rows x 64 k / 128 threads
template <> struct WsRows<true> { u32x4v v[8]; uint32_t hn; };         // bf16 rows: 8 elements per 16-byte load, half as many loads
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
// -> f16 in LDS.  f32 rows: round to nearest even.  bf16 rows: bf16 -> f32 is a shift and f32 -> f16 is then EXACT for
// every value in f16's normal range (8 significant bits fit 11), so a bf16 index carries no row rounding error at all.
template <bool kBf16, bool kL2>
__device__ __forceinline__ void ws_rows_store(_Float16 *buf, uint32_t *hn_buf, uint32_t t, const WsRows<kBf16> &s) {
  if constexpr (kL2) hn_buf[t] = s.hn;
  if constexpr (kBf16) {
    _Float16 *dst = buf + (t >> 3) * kFAStride + (t & 7) * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f16x8 h;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        h[2 * w] = (_Float16)__uint_as_float(s.v[u][w] << 16);
        h[2 * w + 1] = (_Float16)__uint_as_float(s.v[u][w] & 0xFFFF0000u);
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

template <bool kBf16, bool kL2, bool kTiming>
__device__ __forceinline__ void flat_filter_body(const FlatFilterArgs &a) {
  extern __shared__ _Float16 lds_a[];
  constexpr uint32_t kBufHalfs = kFTileRows * kFAStride;                    // one A stage
  uint4 *lds_b = reinterpret_cast<uint4 *>(lds_a + 2 * kBufHalfs);          // [2][kWsBStage]
  uint32_t *lds_ring = reinterpret_cast<uint32_t *>(lds_b + 2 * kWsBStage); // [4 consumer waves][2][64]
  uint32_t *hn_lds = lds_ring + 4 * 2 * kWave;                              // [2][128] (kL2)
  // [2]: cancellation seen during tile T -> word T & 1.  (An LDS-qualified pointer: through a generic one the accesses
  // become FLAT instructions, whose out-of-order return forces every later wait for a load to be vmcnt(0).)
  volatile __attribute__((address_space(3))) uint32_t *lds_stop =
      (volatile __attribute__((address_space(3))) uint32_t *)(hn_lds + 2 * kFTileRows);
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t stages = a.row_stride_f / kFStageK;

  const uint32_t n_tiles = (a.n_rows + kFTileRows - 1) / kFTileRows;
  const uint32_t t_base = n_tiles / gridDim.x, t_rem = n_tiles % gridDim.x;
  const uint32_t first_tile = blockIdx.x * t_base + (blockIdx.x < t_rem ? blockIdx.x : t_rem);
  const uint32_t my_tiles = t_base + (blockIdx.x < t_rem ? 1u : 0u);
  if (my_tiles == 0) return;
  const uint32_t total = my_tiles * stages;
  uint32_t st_c = 0, tile_c = 0, left_c = total;
  bool stop = false;
  if (tid < 2) lds_stop[tid] = 0;
  // the hand-over to the exact kernel was already requested (inputs outside f16, an earlier launch of this batch
  // overflowed): nothing this pass finds would be used.  One thread looks, so that all eight waves agree.
  if (tid == 0) lds_stop[2] = __hip_atomic_load(a.ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (lds_stop[2] != 0) return;

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
  if (wave >= 6) {
    // ================================ query producer =====================================================
    // iteration S: request B(S+2) into the register set B(S) left, write B(S+1) (requested one iteration ago) to slot
    // (S+1) & 1, barrier.  Past the end of the stream the loads re-read its last stage (no branch around a load).
    const uint32_t p = wave - 6;
    FPos lb{first_tile * kFTileRows, 0, total};
    WsB b0, b1;
    ws_b_load(b0, a, p, lb.st, lane);
    fpos_advance(lb, stages);
    ws_b_load(b1, a, p, lb.st, lane);
    fpos_advance(lb, stages);
    ws_b_store(lds_b, p, lane, b0);
    VK_WS_PBARRIER()
    unsigned long long ph[2] = {0, 0}, tlast = __builtin_readcyclecounter();
#define VK_WS_BPROD(PAR, BLOAD, BSTORE)                                                                             \
    {                                                                                                               \
      const bool live = left_c != 0;                                                                                \
      VK_WS_TICK(1)                                                                                                 \
      ws_b_load(BLOAD, a, p, lb.st, lane);                                                                          \
      fpos_advance(lb, stages);                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      ws_b_store(lds_b + ((PAR) ^ 1) * kWsBStage, p, lane, BSTORE);                                                 \
      if constexpr (timing) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
      VK_WS_TICK(0)                                                                                                 \
      left_c -= live ? 1u : 0u;                                                                                     \
      st_c += 1;                                                                                                    \
      if (live && st_c == stages) VK_WS_TILE_END(VK_WS_PBARRIER()) else VK_WS_PBARRIER();                           \
    }
    while (left_c != 0 && !stop) {
      VK_WS_BPROD(0, b0, b1)
      if (stop) break;
      VK_WS_BPROD(1, b1, b0)
    }
#undef VK_WS_BPROD
    if constexpr (timing) {
      if (lane == 0 && a.dbg) {
        atomicAdd(&a.dbg[4], ph[0]);
        atomicAdd(&a.dbg[8], ph[1]);
      }
    }
    return;
  }
  if (wave >= 4) {
    // ================================ row producer =======================================================
    // Rows of stage s live in register set s % 3.  Iteration S: request the rows of stage S+3 into the set stage S
    // left (converted one iteration ago), convert stage S+1 (the 32 loads of S+2 and S+3 stay outstanding) into LDS
    // buffer (S+1) & 1, barrier.  Three stages of rows (96 KB per CU) are in flight.
    const uint32_t t = tid - 256;
    const uint32_t voff = kBf16 ? ((t >> 3) * a.row_stride_f + (t & 7) * 8) * 2u : ((t >> 4) * a.row_stride_f + (t & 15) * 4) * 4u;
    FPos ld{first_tile * kFTileRows, 0, total};
    WsRows<kBf16> x0, x1, x2;
    x0.hn = x1.hn = x2.hn = 0;
    ws_rows_load<kBf16, kL2>(x0, a, ld.row0, ld.st, t, voff);
    fpos_advance(ld, stages);
    ws_rows_load<kBf16, kL2>(x1, a, ld.row0, ld.st, t, voff);
    fpos_advance(ld, stages);
    ws_rows_load<kBf16, kL2>(x2, a, ld.row0, ld.st, t, voff);
    fpos_advance(ld, stages);
    ws_rows_store<kBf16, kL2>(lds_a, hn_lds, t, x0);
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
      ws_rows_load<kBf16, kL2>(RLOAD, a, ld.row0, ld.st, t, voff);                                                  \
      fpos_advance(ld, stages);                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      VK_WS_TICK(0)                                                                                                 \
      ppar ^= 1;                                                                                                    \
      ws_rows_store<kBf16, kL2>(lds_a + ppar * kBufHalfs, hn_lds + ppar * kFTileRows, t, RSTORE);                   \
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
      VK_WS_PROD(x2, x0)
    }
#undef VK_WS_PROD
    if constexpr (timing) {
      if (lane == 0 && a.dbg)
        for (int i = 0; i < 4; ++i) atomicAdd(&a.dbg[i], ph[i]);
    }
    return;
  }
#undef VK_WS_PBARRIER

  // ================================== consumer ===========================================================
  const uint32_t li = lane & 31, g = lane >> 5;
  SurvivorRing ring;
  ring.q = lds_ring + wave * 2 * kWave;
  ring.row = ring.q + kWave;
  ring.cnt = 0;
  const bool has_q = wave * 2 < a.nqt;
  float thr[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) thr[t2] = wave * 2 + t2 < a.nqt ? a.thr[(wave * 2 + t2) * 32 + li] : __builtin_inff();
  f32x16 acc[2][4];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[t2][rt] = zero;
  uint32_t tile_row0 = first_tile * kFTileRows;
  uint32_t par = 0;
  __syncthreads();                                            // (the prologue's barrier)

  // iteration S: 32 MFMAs of stage S, operands from LDS (A: buffer S & 1, row li of each row tile; B: slot S & 1,
  // query tiles 2w, 2w+1), fetched one K-step ahead; gate at the end of a tile; barrier.  Not unrolled (the buffer and
  // the slot are offsets in a register): one copy of the gate's code.
#define VK_WS_AB(FA, FB0, FB1, KK)                                                                                  \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                  \
    FA[rt] = *reinterpret_cast<const f16x8 *>(ab + rt * 32 * kFAStride + (KK) * 16);                                \
  FB0 = bb[(KK) * kWave];                                                                                           \
  FB1 = bb[(4 + (KK)) * kWave];
#define VK_WS_MM(FA, FB0, FB1)                                                                                      \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                                \
    acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[rt], FB0, acc[0][rt], 0, 0, 0);                          \
    acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[rt], FB1, acc[1][rt], 0, 0, 0);                          \
  }
  // (first K-step of a tile: the accumulators are not cleared -- 128 register moves -- but started from the constant 0)
#define VK_WS_MMZ(FA, FB0, FB1)                                                                                     \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                                \
    acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[rt], FB0, zero, 0, 0, 0);                                \
    acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[rt], FB1, zero, 0, 0, 0);                                \
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
  unsigned long long ph[3] = {0, 0, 0}, tlast = __builtin_readcyclecounter();
  while (left_c != 0 && !stop) {
    VK_WS_TICK(2)
    const _Float16 *ab = lds_a + par * kBufHalfs + li * kFAStride + g * 8;
    const f16x8 *bb = reinterpret_cast<const f16x8 *>(lds_b + par * kWsBStage + (wave * 2) * 4 * kWave) + lane;
    if (has_q) {
      if (st_c == 0) VK_WS_STAGE(VK_WS_MMZ) else VK_WS_STAGE(VK_WS_MM)
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
    if (st_c == stages) {
      if (has_q) {
        filter_gate<4, false>(a, acc[0], thr[0], tile_row0, wave * 2, li, g, ring, lane);
        filter_gate<4, false>(a, acc[1], thr[1], tile_row0, wave * 2 + 1, li, g, ring, lane);
      }
      tile_row0 += kFTileRows;
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
    if (lane == 0 && a.dbg)
      for (int i = 0; i < 3; ++i) atomicAdd(&a.dbg[5 + i], ph[i]);
  }
  ring_flush(a, ring, lane);
}

template <bool kBf16, bool kL2, bool kTiming>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, kTiming>(a);
}
// the same kernel under another name for the pass over the bound's sample (a few per cent of the rows), so that a
// profile's per-kernel averages are averages over launches of one size
template <bool kBf16, bool kL2>
__global__ __launch_bounds__(kWsThreads, 1) void flat_filter_sample_kernel(FlatFilterArgs a) {
  flat_filter_body<kBf16, kL2, false>(a);
}

size_t flat_filter_lds_bytes() {
  return (size_t)2 * kFTileRows * kFAStride * sizeof(_Float16) + (size_t)2 * kWsBStage * 16 + (size_t)4 * 2 * kWave * 4 +
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
  if (a.nqt == 0 || a.nqt > 8 || blocks == 0) return hipErrorInvalidValue;
  const size_t lds = flat_filter_lds_bytes();
  const void *fn = a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_kernel<true, true, false>)
                                  : reinterpret_cast<const void *>(&flat_filter_kernel<true, false, false>))
                          : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_kernel<false, true, false>)
                                  : reinterpret_cast<const void *>(&flat_filter_kernel<false, false, false>));
  if (a.sample_pass)
    fn = a.bf16 ? (a.l2 ? reinterpret_cast<const void *>(&flat_filter_sample_kernel<true, true>)
                        : reinterpret_cast<const void *>(&flat_filter_sample_kernel<true, false>))
                : (a.l2 ? reinterpret_cast<const void *>(&flat_filter_sample_kernel<false, true>)
                        : reinterpret_cast<const void *>(&flat_filter_sample_kernel<false, false>));
  if (a.timing) {
    if (a.bf16 || a.l2) return hipErrorInvalidValue;
    fn = reinterpret_cast<const void *>(&flat_filter_kernel<false, false, true>);
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // (above the 64 KB default)
  if (e != hipSuccess) return e;
  FlatFilterArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(fn, dim3(blocks), dim3(kWsThreads), params, lds, s);
}

}  // namespace vk
