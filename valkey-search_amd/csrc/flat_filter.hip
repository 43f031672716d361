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
    // no bound yet (fewer than k allowed rows in the sample) or inputs that cannot go through f16: every row passes,
    // the list overflows, and the exact kernel answers this batch
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

// ---- the filter ------------------------------------------------------------------------------------------------------
// Block = 512 threads = 8 waves, two per SIMD (so that one wave's wait for memory is the other's matrix time): wave w
// owns query tile w (32 queries) against all 128 rows of the tile = four 32 x 32 accumulator tiles.
constexpr int kFThreads = 512;
// this thread's share of one stage: 128 rows x 64 k = 2048 groups of 4 elements / 512 threads (16 B of f32, 8 B of bf16)
template <bool kBf16> struct RowStage { float4 v[4]; uint32_t hn; };
template <> struct RowStage<true> { uint2 v[4]; uint32_t hn; };

// (kL2: every stage also carries the packed half norm of row tid % 128 of its tile -- 4 bytes per thread, L2 hits after
// the tile's first stage -- so that it travels through the same register sets and LDS buffers as the rows, without a
// load or a branch of its own in the loop)
template <bool kBf16, bool kL2>
__device__ __forceinline__ RowStage<kBf16> stage_rows_load(const FlatFilterArgs &a, uint32_t tile_row0, uint32_t st, uint32_t tid) {
  RowStage<kBf16> s;
  s.hn = 0;
  if constexpr (kL2) {
    const uint32_t r = tile_row0 + (tid & (kFTileRows - 1));
    s.hn = a.hn16[r < a.n_rows ? r : a.n_rows - 1];
  }
  // idx = tid + 512 u: row = idx / 16, 4-element column idx % 16 of the row's 64-element stage slice (coalesced per row)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t idx = tid + (uint32_t)kFThreads * u;
    const size_t e = ((size_t)tile_row0 + (idx >> 4)) * a.row_stride_f + st * kFStageK + (idx & 15) * 4;
    if constexpr (kBf16) s.v[u] = *reinterpret_cast<const uint2 *>(static_cast<const uint16_t *>(a.rows) + e);
    else s.v[u] = *reinterpret_cast<const float4 *>(static_cast<const float *>(a.rows) + e);
  }
  return s;
}

// -> f16 in LDS.  f32 rows: round to nearest even.  bf16 rows: bf16 -> f32 is a shift and f32 -> f16 is then EXACT for
// every value in f16's normal range (8 significant bits fit 11), so a bf16 index carries no row rounding error at all.
template <bool kBf16, bool kL2>
__device__ __forceinline__ void stage_rows_store(_Float16 *buf, uint32_t *hn_buf, uint32_t tid, const RowStage<kBf16> &s) {
  if constexpr (kL2) {
    if (tid < (uint32_t)kFTileRows) hn_buf[tid] = s.hn;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t idx = tid + (uint32_t)kFThreads * u;
    f16x4 h;
    if constexpr (kBf16) {
      h[0] = (_Float16)__uint_as_float(s.v[u].x << 16);
      h[1] = (_Float16)__uint_as_float(s.v[u].x & 0xFFFF0000u);
      h[2] = (_Float16)__uint_as_float(s.v[u].y << 16);
      h[3] = (_Float16)__uint_as_float(s.v[u].y & 0xFFFF0000u);
    } else {
      h[0] = (_Float16)s.v[u].x;
      h[1] = (_Float16)s.v[u].y;
      h[2] = (_Float16)s.v[u].z;
      h[3] = (_Float16)s.v[u].w;
    }
    *reinterpret_cast<f16x4 *>(buf + (idx >> 4) * kFAStride + (idx & 15) * 4) = h;
  }
}

struct BFrags { f16x8 b[4]; };   // this wave's query tile x the stage's four K-steps

__device__ __forceinline__ BFrags stage_b_load(const FlatFilterArgs &a, uint32_t wave, uint32_t st, uint32_t lane) {
  BFrags f;
  const uint32_t ks_n = a.row_stride_f / 16;
  const uint32_t jt = wave < a.nqt ? wave : a.nqt - 1;   // (a wave without queries re-reads the last tile)
  const f16x8 *p = reinterpret_cast<const f16x8 *>(a.q16) + ((size_t)(jt * ks_n + st * 4) * kWave + lane);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f.b[kk] = p[(size_t)kk * kWave];
  return f;
}

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
template <bool kTiny = false>
__device__ __forceinline__ void filter_gate(const FlatFilterArgs &a, f32x16 (&acc)[4], float thr, uint32_t tile_row0,
                                            uint32_t wave, uint32_t li, uint32_t g, SurvivorRing &ring, uint32_t lane) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    float m = acc[rt][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rt][r]);
    if (__builtin_amdgcn_ballot_w64(m >= thr) != 0) {   // (rare: a survivor somewhere in this 32 x 32 block)
      if constexpr (kTiny) { ring.cnt += 1; acc[rt] = zero; continue; }
      const uint32_t q = wave * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = tile_row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        bool pass = acc[rt][r] >= thr && row < a.n_rows && q < a.nq;
        if (pass && a.allow_bits != nullptr) pass = allow_bit(a.allow_bits, a.allow_nbits, a.labels[row]);
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
    acc[rt] = zero;
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

// kAblate: timing experiments only (compile-time, so that the product kernel has no branches around its loads)
template <int kAblate, bool kBf16, bool kL2>
__global__ __launch_bounds__(512, 1) void flat_filter_kernel(FlatFilterArgs a) {
  extern __shared__ _Float16 lds_a[];   // [2 stages][128 rows][kFAStride], then the waves' survivor rings
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t li = lane & 31, g = lane >> 5;
  const uint32_t stages = a.row_stride_f / kFStageK;
  constexpr uint32_t kBufHalfs = kFTileRows * kFAStride;
  SurvivorRing ring;
  ring.q = reinterpret_cast<uint32_t *>(lds_a + 2 * kBufHalfs) + wave * 2 * kWave;
  ring.row = ring.q + kWave;
  ring.cnt = 0;
  uint32_t *hn_lds = reinterpret_cast<uint32_t *>(lds_a + 2 * kBufHalfs) + (kFThreads / kWave) * 2 * kWave;   // [2][128] (kL2)

  // this block's contiguous range of row tiles
  const uint32_t n_tiles = (a.n_rows + kFTileRows - 1) / kFTileRows;
  const uint32_t t_base = n_tiles / gridDim.x, t_rem = n_tiles % gridDim.x;
  const uint32_t first_tile = blockIdx.x * t_base + (blockIdx.x < t_rem ? blockIdx.x : t_rem);
  const uint32_t my_tiles = t_base + (blockIdx.x < t_rem ? 1u : 0u);
  if (my_tiles == 0) return;
  const uint32_t total = my_tiles * stages;
  const bool has_q = wave < a.nqt;
  const float thr = has_q ? a.thr[wave * 32 + li] : __builtin_inff();   // gate of this lane's query column

  // Software pipeline, iteration S = stage S of the stream:
  //   top     the B operands (L2) and the rows (HBM) of stage S+4 into the register sets whose previous content (stage S)
  //           is consumed in this iteration / went to LDS one iteration ago.  Both run the SAME distance ahead: loads
  //           return in order (one vmcnt counter), so a B operand fetched one iteration ahead would make the wait for it
  //           a wait for every row load issued before it -- the rows would have one iteration of cover, not four.
  //           Three to four stages of rows (96-128 KB per CU) are in flight: at 6 TB/s the loaded HBM latency is
  //           several microseconds
  //   middle  the 16 MFMAs of stage S: A fragments from LDS buffer S & 1, fetched one K-step ahead of their use
  //   bottom  stage S+1 (loaded one iteration ago) converted to f16 into the other LDS buffer, one barrier
  // Unrolled by four with the register sets named explicitly (rows and B operands of stage s in sets s % 4):
  // rotating them through a copy would make the copy wait for loads that are still in flight.
  FPos ld{first_tile * kFTileRows, 0, total};   // next stage whose rows are fetched
  FPos lb = ld;                                  // next stage whose B operands are fetched
  RowStage<kBf16> x0 = stage_rows_load<kBf16, kL2>(a, ld.row0, ld.st, tid), x1, x2, x3;
  stage_rows_store<kBf16, kL2>(lds_a, hn_lds, tid, x0);
  BFrags b0 = stage_b_load(a, wave, lb.st, lane), b1, b2, b3;
  fpos_advance(ld, stages);
  fpos_advance(lb, stages);
  b1 = stage_b_load(a, wave, lb.st, lane);        // stages 1, 2, 3: B operands, then rows
  x1 = stage_rows_load<kBf16, kL2>(a, ld.row0, ld.st, tid);
  fpos_advance(ld, stages);
  fpos_advance(lb, stages);
  b2 = stage_b_load(a, wave, lb.st, lane);
  x2 = stage_rows_load<kBf16, kL2>(a, ld.row0, ld.st, tid);
  fpos_advance(ld, stages);
  fpos_advance(lb, stages);
  b3 = stage_b_load(a, wave, lb.st, lane);
  x3 = stage_rows_load<kBf16, kL2>(a, ld.row0, ld.st, tid);
  fpos_advance(ld, stages);
  fpos_advance(lb, stages);
  __syncthreads();

  f32x16 acc[4];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = zero;

  uint32_t tile_row0 = first_tile * kFTileRows;
  uint32_t st_c = 0, left_c = total, tile_c = 0;
  uint32_t cancel_now = 0;
  bool stop = false;
  const uint32_t hot_row0 = first_tile * kFTileRows;
  // (kAblate & 128: cycles per phase -- 0 fragment reads + MFMAs, 1 issue of the loads, 2 wait for the rows + convert +
  // LDS stores, 3 gate, 4 barrier -- summed over the waves into a.dbg)
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();

#define VK_FMMA(AV, KK)                                                                                             \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                  \
    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AV[rt], bcur_.b[KK], acc[rt], 0, 0, 0);
#define VK_FAREAD(AV, KK)                                                                                           \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                  \
    AV[rt] = *reinterpret_cast<const f16x8 *>(ab + rt * 32 * kFAStride + (KK) * 16);

#define VK_TICK(I)                                                                                                  \
  if constexpr (kAblate & 128) {                                                                                    \
    const unsigned long long now_ = __builtin_readcyclecounter();                                                   \
    ph[I] += now_ - tlast;                                                                                          \
    tlast = now_;                                                                                                   \
  }
#define VK_FSTAGE(PAR, RLOAD, RSTORE, BCUR, BNEXT)                                                                  \
  {                                                                                                                 \
    VK_TICK(4)                                                                                                      \
    const bool live = left_c != 0;                                                                                  \
    if (live && st_c == 0 && a.cancel && (tile_c % kCancelPollTiles) == 0)                                          \
      cancel_now = __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                        \
    const _Float16 *ab = lds_a + (PAR) * kBufHalfs + li * kFAStride + g * 8;                                        \
    f16x8 fa[4], fb[4];                                                                                             \
    VK_FAREAD(fa, 0)                                                                                                \
    const BFrags bcur_ = BCUR;                                                                                      \
    if (has_q && live && !(kAblate & 4)) {                                                                          \
      VK_FAREAD(fb, 1)                                                                                              \
      VK_FMMA(fa, 0)                                                                                                \
      VK_FAREAD(fa, 2)                                                                                              \
      VK_FMMA(fb, 1)                                                                                                \
      VK_FAREAD(fb, 3)                                                                                              \
      VK_FMMA(fa, 2)                                                                                                \
      VK_FMMA(fb, 3)                                                                                                \
      /* operands of a K-step are requested a whole K-step (four MFMAs) before their use */                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                            \
      if constexpr (kL2) {                                                                                          \
        if (st_c + 1 == stages) {   /* one more K-step: (hn_hi, hn_lo, 0 ...) x (-1, -1, 0 ...) = - |x|^2 / 2 */    \
          const uint4 nb = make_uint4(g == 0 ? 0xBC00BC00u : 0u, 0u, 0u, 0u);                                       \
          _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) {                                                        \
            const uint4 na = make_uint4(g == 0 ? hn_lds[(PAR) * kFTileRows + rt * 32 + li] : 0u, 0u, 0u, 0u);       \
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, na), __builtin_bit_cast(f16x8, nb), \
                                                             acc[rt], 0, 0, 0);                                     \
          }                                                                                                         \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    if constexpr (kAblate & 128) asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");            \
    VK_TICK(0)                                                                                                      \
    /* (the loads go out behind the MFMAs: placed between them -- one per MFMA -- the f32 kernel lost 40 %: the  */ \
    /* compiler then also hoists the LDS stores, and with them the wait for HBM data, into the MFMA sequence)   */ \
    if constexpr (!(kAblate & 16)) BNEXT = stage_b_load(a, wave, lb.st, lane);                                      \
    fpos_advance(lb, stages);                                                                                       \
    if constexpr (!(kAblate & 32)) RLOAD = stage_rows_load<kBf16, kL2>(a, (kAblate & 1) ? hot_row0 : ld.row0, ld.st, tid); \
    fpos_advance(ld, stages);                                                                                       \
    VK_TICK(1)                                                                                                      \
    if constexpr (!(kAblate & 8)) stage_rows_store<kBf16, kL2>(lds_a + ((PAR) ^ 1) * kBufHalfs, hn_lds + ((PAR) ^ 1) * kFTileRows, tid, RSTORE); \
    if constexpr (kAblate & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
    VK_TICK(2)                                                                                                      \
    left_c -= live ? 1u : 0u;                                                                                       \
    st_c += 1;                                                                                                      \
    if (live && st_c == stages) {                                                                                   \
      if (has_q && !(kAblate & 2)) filter_gate<(kAblate & 64) != 0>(a, acc, thr, tile_row0, wave, li, g, ring, lane);                                \
      st_c = 0;                                                                                                     \
      tile_c += 1;                                                                                                  \
      tile_row0 += kFTileRows;                                                                                      \
      VK_TICK(3)                                                                                                    \
      stop = __syncthreads_or((int)cancel_now) != 0;   /* block-uniform: every wave leaves at the same tile */      \
    } else {                                                                                                        \
      __syncthreads();                                                                                              \
    }                                                                                                               \
  }

  while (left_c != 0 && !stop) {
    VK_FSTAGE(0, x0, x1, b0, b0)
    if (stop) break;
    VK_FSTAGE(1, x1, x2, b1, b1)
    if (stop) break;
    VK_FSTAGE(0, x2, x3, b2, b2)
    if (stop) break;
    VK_FSTAGE(1, x3, x0, b3, b3)
  }
#undef VK_FSTAGE
#undef VK_TICK
#undef VK_FMMA
#undef VK_FAREAD
  if constexpr (kAblate & 128) {
    if (lane == 0 && a.dbg)
      for (int i = 0; i < 5; ++i) atomicAdd(&a.dbg[i], ph[i]);
  }
  if constexpr (kAblate & 64) { if (ring.cnt == 0xFFFFFFFFu) a.ovf[0] = 1; return; }
  ring_flush(a, ring, lane);
}

size_t flat_filter_lds_bytes() {
  return (size_t)2 * kFTileRows * kFAStride * sizeof(_Float16) + (size_t)(kFThreads / kWave) * 2 * kWave * 4 + (size_t)2 * kFTileRows * 4;
}

bool flat_filter_supported(uint32_t row_stride_f, uint64_t k, bool bf16, bool l2) {
  (void)bf16;
  (void)l2;
  return (row_stride_f % kFStageK) == 0 && k >= 1 && k <= 64;
}

hipError_t launch_flat_qprep(const FlatFilterArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(flat_qprep_kernel, dim3(a.nqt * 32), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_flat_filter(const FlatFilterArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.nqt == 0 || a.nqt > 8 || blocks == 0) return hipErrorInvalidValue;
  const dim3 grid(blocks), block(kFThreads);
  const size_t lds = flat_filter_lds_bytes();
  if (a.bf16 || a.l2) {
    if (a.ablate) return hipErrorInvalidValue;
    if (a.bf16 && a.l2) hipLaunchKernelGGL((flat_filter_kernel<0, true, true>), grid, block, lds, s, a);
    else if (a.bf16) hipLaunchKernelGGL((flat_filter_kernel<0, true, false>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((flat_filter_kernel<0, false, true>), grid, block, lds, s, a);
    return hipGetLastError();
  }
  switch (a.ablate) {
    case 0: hipLaunchKernelGGL((flat_filter_kernel<0, false, false>), grid, block, lds, s, a); break;
    case 128: hipLaunchKernelGGL((flat_filter_kernel<128, false, false>), grid, block, lds, s, a); break;
    case 64: hipLaunchKernelGGL((flat_filter_kernel<64, false, false>), grid, block, lds, s, a); break;
    case 32: hipLaunchKernelGGL((flat_filter_kernel<32, false, false>), grid, block, lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace vk
