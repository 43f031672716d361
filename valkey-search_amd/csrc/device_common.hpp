// device_common.hpp -- CDNA4 (gfx950) device building blocks shared by the FLAT scan,
// the label gather-scan, the HNSW expansion and the top-k merge kernels.
//
// Distance arithmetic.  The reference's fstdistfunc_ is SimSIMD 5.0.1's f32 dot / l2sq
// as dispatched on an AVX-512 host (third_party/simsimd/include/simsimd/dot.h:1183-1204,
// spatial.h:1131-1153): 16 independent f32 accumulators, accumulator l fed
// fma(a[16c+l], b[16c+l], acc[l]) for c = 0,1,..., zero-masked tail, then the
// _mm512_reduce_add_ps tree (l,l+8) -> (l,l+4) -> (l,l+2) -> (0,1); the hnswlib bridge
// returns (float)(1.0 - (double)dot) resp. (float)l2sq (third_party/hnswlib/simsimd.h:16-34).
// To return bit-identical distances (hence identical neighbour ids, ties included) the
// device keeps exactly those 16 chains: a row is owned by a QUAD of lanes, lane j of
// the quad holds accumulators 4j..4j+3 and walks the row 64 B (one chunk of 16 floats)
// at a time with one 16-B load per lane; the tree is two DPP quad permutes plus three
// adds.  Rows are stored zero-padded to a multiple of 16 floats, which reproduces the
// masked tail (fma(0,0,x) == x; an accumulator that starts at +0 never becomes -0).
// 1.0f - dot equals the reference's double-precision subtract: a single add of two
// f32 values rounded through f64 is innocuous double rounding (53 >= 2*24+2).
//
// Everything here must be compiled with -ffp-contract=off: only the explicit fmaf may fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vk {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kRowsPerWave = 16;     // one row per quad of lanes
constexpr uint64_t kNoLabel = ~0ull;

__device__ __forceinline__ float dpp_quad_xor1(float v) {
  // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
  // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// Wave-wide maximum, wave-uniform result, without the LDS crossbar: butterfly inside each row of 16 lanes with DPP
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row results through scalar registers.
// (__shfl_xor is a ds_bpermute per step; this chain is what a top-k update waits on.)
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_quad_xor1(v));
  v = fmaxf(v, dpp_quad_xor2(v));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));  // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float wave_min_f32(float v) { return -wave_max_f32(-v); }

// acc = this lane's accumulators 4j..4j+3 of a row.  Every lane of the quad returns the
// full 16-accumulator sum in _mm512_reduce_add_ps order.
__device__ __forceinline__ float quad_reduce16(float4 acc) {
  acc.x += dpp_quad_xor2(acc.x);  // (l, l+8): lane j <-> lane j^2
  acc.y += dpp_quad_xor2(acc.y);
  acc.z += dpp_quad_xor2(acc.z);
  acc.w += dpp_quad_xor2(acc.w);
  acc.x += dpp_quad_xor1(acc.x);  // (l, l+4): lane j <-> lane j^1
  acc.y += dpp_quad_xor1(acc.y);
  acc.z += dpp_quad_xor1(acc.z);
  acc.w += dpp_quad_xor1(acc.w);
  float u0 = acc.x + acc.z;       // (l, l+2)
  float u1 = acc.y + acc.w;
  return u0 + u1;                 // (0, 1)
}

template <bool kL2>
__device__ __forceinline__ void chunk_fma(float4 &acc, const float4 x, const float4 q) {
  if constexpr (kL2) {
    float dx = x.x - q.x, dy = x.y - q.y, dz = x.z - q.z, dw = x.w - q.w;
    acc.x = fmaf(dx, dx, acc.x);
    acc.y = fmaf(dy, dy, acc.y);
    acc.z = fmaf(dz, dz, acc.z);
    acc.w = fmaf(dw, dw, acc.w);
  } else {
    acc.x = fmaf(x.x, q.x, acc.x);
    acc.y = fmaf(x.y, q.y, acc.y);
    acc.z = fmaf(x.z, q.z, acc.z);
    acc.w = fmaf(x.w, q.w, acc.w);
  }
}

template <bool kL2>
__device__ __forceinline__ float finish_distance(float sum) {
  if constexpr (kL2) return sum;   // (float)l2sq: exact
  return 1.0f - sum;               // == (float)(1.0 - (double)dot), see header comment
}

// ---- row storage type ------------------------------------------------------------------------
// Rows are f32 (the only type the reference has, vector_base.h:112-114) or bf16 (extension, config 4):
// bf16 rows are the f32 input rounded to nearest-even at ingest; the arithmetic stays the f32
// lane-exact chain on the widened values (bf16 -> f32 is a 16-bit shift, exact), so a bf16 index
// answers exactly like an f32 index over the rounded rows.  4-element "pieces": piece i of a row
// is elements 4i..4i+3 (16 B of f32, 8 B of bf16).
template <bool kBf16>
__device__ __forceinline__ const char *row_base(const void *rows, size_t row, uint32_t stride_e) {
  return static_cast<const char *>(rows) + row * (size_t)stride_e * (kBf16 ? 2 : 4);
}
template <bool kBf16>
__device__ __forceinline__ float4 row_piece(const char *base, uint32_t piece) {
  if constexpr (kBf16) {
    const uint2 u = reinterpret_cast<const uint2 *>(base)[piece];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
  } else {
    return reinterpret_cast<const float4 *>(base)[piece];
  }
}

// One row against the query block in LDS, by one quad (all 4 lanes return the distance).
// base = start of the row; qs = padded query as float4[chunks*4]; j = lane inside the quad.
// kBatch = 16-B pieces in flight per lane before the first one is consumed.  8 keeps the register
// count low (throughput kernels, many waves per SIMD); the latency-bound HNSW search of a few queries
// uses 24, so a 768-d row costs two dependent memory round trips instead of six.
template <bool kL2, bool kBf16, int kBatch = 8>
__device__ __forceinline__ float quad_row_distance(const char *__restrict__ base, const float4 *qs, uint32_t chunks, int j) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t c = 0;
  for (; c + kBatch <= chunks; c += kBatch) {
    float4 x[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
    for (int u = 0; u < kBatch; ++u) chunk_fma<kL2>(acc, x[u], qs[(c + u) * 4 + j]);
  }
  if constexpr (kBatch > 8) {
    for (; c + 8 <= chunks; c += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
      for (int u = 0; u < 8; ++u) chunk_fma<kL2>(acc, x[u], qs[(c + u) * 4 + j]);
    }
  }
  for (; c < chunks; ++c) chunk_fma<kL2>(acc, row_piece<kBf16>(base, c * 4 + j), qs[c * 4 + j]);
  return finish_distance<kL2>(quad_reduce16(acc));
}

// The same distance for FEWER rows than the wave has quads: kSplit (2 or 4) quads share a row.  Quad g of a row
// loads batch s*kSplit+g of its pieces while the others load theirs, then every quad receives the batches in
// order through the cross-lane network and accumulates them, so the fma chain per lane class is the one of
// quad_row_distance (bit-identical result) while one memory round trip covers kSplit batches.  Lane layout: group
// g = lane / (64/kSplit); inside a group the usual quads (row = (lane % (64/kSplit)) / 4, j = lane % 4); `base` must
// be the same row in every group.  All lanes return the distance.
template <bool kL2, bool kBf16, int kBatch, int kSplit>
__device__ __forceinline__ float quad_row_distance_split(const char *__restrict__ base, const float4 *qs, uint32_t chunks, int lane) {
  constexpr int kGroupLanes = kWave / kSplit;
  const int j = lane & 3, g = lane / kGroupLanes, in_group = lane % kGroupLanes;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t batches = chunks / kBatch;          // the caller checked chunks % kBatch == 0
  for (uint32_t s0 = 0; s0 < batches; s0 += kSplit) {
    float4 x[kBatch];
    const uint32_t mine = s0 + g < batches ? s0 + g : batches - 1;   // (a group past the end re-reads the last batch)
#pragma unroll
    for (int u = 0; u < kBatch; ++u) x[u] = row_piece<kBf16>(base, (mine * kBatch + u) * 4 + j);
#pragma unroll
    for (int t = 0; t < kSplit; ++t) {
      if (s0 + t < batches) {
        const int src = t * kGroupLanes + in_group;
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const float4 y = make_float4(__shfl(x[u].x, src), __shfl(x[u].y, src), __shfl(x[u].z, src), __shfl(x[u].w, src));
          chunk_fma<kL2>(acc, y, qs[((s0 + t) * kBatch + u) * 4 + j]);
        }
      }
    }
  }
  return finish_distance<kL2>(quad_reduce16(acc));
}

// ---- (distance,label) total order: std::pair<float,size_t> operator< ------------------
__device__ __forceinline__ bool dl_less(float da, uint64_t la, float db, uint64_t lb) {
  return da < db || (da == db && la < lb);
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  uint32_t lo = __shfl_xor((int)(uint32_t)v, m), hi = __shfl_xor((int)(uint32_t)(v >> 32), m);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f32(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- per-wave running top-k ---------------------------------------------------------------
// The k best (smallest by (distance,label)) entries seen by one wave, spread over the
// lanes: slot s = e*64 + lane, kE slots per lane, k <= 64*kE.  The current worst kept
// entry (the heap top of bruteforce.h:128-141) is cached wave-uniformly so the hot loop
// pays one compare per row; the rare insert replaces the worst slot and re-reduces.
template <int kE>
struct WaveTopK {
  float d[kE];
  uint64_t lab[kE];
  uint32_t k, cnt;       // wave-uniform
  float thr_d;           // wave-uniform: worst kept distance, +inf while cnt < k
  uint64_t thr_lab;
  uint32_t thr_slot;

  __device__ __forceinline__ void init(uint32_t k_) {
    k = k_;
    cnt = 0;
    thr_d = __builtin_inff();
    thr_lab = kNoLabel;
    thr_slot = 0;
#pragma unroll
    for (int e = 0; e < kE; ++e) { d[e] = __builtin_inff(); lab[e] = kNoLabel; }
  }

  __device__ __forceinline__ void recompute_worst(int lane) {
    float bd = -__builtin_inff();
    uint64_t bl = 0;
    uint32_t bs = 0;
    bool have = false;
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      uint32_t s = (uint32_t)e * kWave + lane;
      if (s < k && (!have || dl_less(bd, bl, d[e], lab[e]))) { bd = d[e]; bl = lab[e]; bs = s; have = true; }
    }
    if (!have) { bd = -__builtin_inff(); bl = 0; }
    // fast path: reduce the distance alone (one shuffle per step instead of five); only when several lanes
    // hold the maximum distance does the label have to take part
    const float md = wave_max_f32(bd);
    const uint64_t at_max = __ballot(have && bd == md);
    if (__popcll(at_max) == 1) {
      const int src = __ffsll((unsigned long long)at_max) - 1;
      thr_d = md;                        // (wave-uniform, in scalar registers)
      thr_lab = readlane_u64(bl, src);
      thr_slot = (uint32_t)__builtin_amdgcn_readlane((int)bs, src);
      return;
    }
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
      float od = __shfl_xor(bd, m);
      uint64_t ol = shfl_xor_u64(bl, m);
      uint32_t os = __shfl_xor((int)bs, m);
      int oh = __shfl_xor((int)have, m);
      if (oh && (!have || dl_less(bd, bl, od, ol))) { bd = od; bl = ol; bs = os; have = true; }
    }
    thr_d = readlane_f32(bd, 0);
    thr_lab = readlane_u64(bl, 0);
    thr_slot = (uint32_t)__builtin_amdgcn_readlane((int)bs, 0);
  }

  // (cd, cl) wave-uniform.  Caller has already checked the distance gate cd <= thr_d.
  __device__ __forceinline__ void insert(float cd, uint64_t cl, int lane) {
    uint32_t slot;
    if (cnt < k) {
      slot = cnt++;
    } else {
      if (!dl_less(cd, cl, thr_d, thr_lab)) return;
      slot = thr_slot;
    }
#pragma unroll
    for (int e = 0; e < kE; ++e)
      if (slot == (uint32_t)e * kWave + lane) { d[e] = cd; lab[e] = cl; }
    if (cnt == k) recompute_worst(lane);
  }

  // Write the kept entries to out[0..k) (unsorted; empty slots are (+inf, kNoLabel)).
  __device__ __forceinline__ void store(float *out_d, uint64_t *out_l, int lane) const {
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      uint32_t s = (uint32_t)e * kWave + lane;
      if (s < k) { out_d[s] = d[e]; out_l[s] = lab[e]; }
    }
  }
};

// the host's cancellation word (pinned host memory, written by the thread that waits for the kernel): a system-scope
// load, one PCIe read per wave and poll
__device__ __forceinline__ bool poll_cancel(const uint32_t *flag) {
  return flag != nullptr && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}

// device-side conditional launch (FlatScanArgs::run_flag): true = this launch has nothing to do
__device__ __forceinline__ bool launch_skipped(const uint32_t *run_flag, uint32_t run_if, uint32_t run_hi) {
  if (run_flag == nullptr) return false;
  const uint32_t v = __hip_atomic_load(run_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return run_hi != 0 ? (v < run_if || v > run_hi) : v != run_if;
}

__device__ __forceinline__ bool allow_bit(const uint64_t *__restrict__ bits, uint64_t nbits, uint64_t label) {
  if (!bits) return true;
  if (label >= nbits) return false;
  return (bits[label >> 6] >> (label & 63)) & 1ull;
}

// ---- candidate filter: what both the gate (flat_filter.hip) and the re-rank's second bound (flat_scan.hip) compute ----
// The bound on |approximate score - the reference's exact score| for a row of norm <= R against a query whose error
// polynomial is co = (c2, c1, c0, state) (flat_qprep_kernel): c2 R^2 + c1 R + c0 (c2 = 0 in the inner-product space).
template <bool kL2> __device__ __forceinline__ float filter_margin(float c2, float c1, float c0, float R) {
  if constexpr (kL2) return fmaf(fmaf(c2, R, c1), R, c0);
  else return fmaf(c1, R, c0);
}
// order-preserving key of a float: larger float <-> larger key (NaN must not get here)
__device__ __forceinline__ uint32_t desc_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float desc_key_float(uint32_t key) {
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}

}  // namespace vk
