// flat_gemm.hip -- K4: batched FLAT search for the inner-product space (IP / COSINE) on the
// gfx950 matrix cores: 1 - Q.X^T with v_mfma_f32_32x32x2_f32 and a fused per-lane top-k,
// BIT-IDENTICAL to the reference CPU distance.
//
// Why it can be exact.  The reference distance is SimSIMD's AVX-512 dot (dot.h:1183-1204): 16
// lane-class accumulators, class l = element index mod 16, each an in-order fmaf chain over the
// chunks c = 0,1,..., then a fixed add tree, then 1.0f - dot (hnswlib/simsimd.h:16-24).  The
// f32-input MFMA is bit-for-bit a k-ordered fmaf chain (no wider internal accumulation), so one
// accumulator tile PER LANE CLASS, fed only (chunk c, class l), (chunk c+1, class l), ... in order,
// reproduces chain l for all 32x32 (row,query) pairs of the tile at once; the 16 tiles are then
// combined elementwise with the same tree.  Cost = the FLOPs of one K=D GEMM.  (L2 is not a
// product of the inputs and stays on the scan kernel.)
//
// Tiling (one 256-thread block per CU, 4 waves = one per SIMD, ~400 VGPR+AGPR per lane):
//   block tile  = 128 rows x 32 queries; wave w owns rows [32w,32w+32) x all 32 queries
//   accumulators= 16 classes x f32x16 = 256 registers per lane
//   Q tile      : resident in LDS for the whole block, [32][Dp+4] f32 (pad 4 => ds_read_b128 hits
//                 64 distinct banks per 16-lane group)
//   X rows      : each wave streams ITS 32 rows HBM -> registers -> its private LDS ring (3 stages of
//                 [32 rows][2 chunks], row stride 36 dwords, conflict free) -> MFMA operands; no
//                 block barrier in the main loop, the waves run free
//   per stage   : for each class group p (classes 4p..4p+3): one ds_read_b128 of X, one of Q
//                 (lane = (row or query i = lane&31, chunk kk = lane>>5)), then 4 MFMAs
//   per tile    : tree-reduce the 16 tiles, 1-dot, compare the lane's 16 (row) values for its query
//                 against the lane's running threshold; the rare insert goes to the lane's private
//                 k-list in HBM scratch ((distance,label) order, ties by label)
// Grid: persistent, nrp x nqt blocks, XCD-aware decode (blocks that stream the same rows for
// different query tiles sit on one XCD and share its L2).
// Roofline: MFMA-bound, 2*rows*D*B FLOPs per launch against the 157.3 TFLOP/s f32 matrix peak.
#include <stdlib.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace vk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kTileRows = 128;
constexpr int kXStride = 36;     // dwords per staged row: 2 chunks (32 floats) + 4 pad
constexpr int kXBufs = 3;
}  // namespace

struct LaneTop {          // one lane's running top-k state; the list itself lives in HBM scratch
  float thr_d;            // worst kept distance, +inf while the list is not full
  uint64_t thr_l;
  uint32_t thr_i, cnt;
};

// order-preserving f32 <-> u32 (larger float = larger key), for atomicMin on the shared bound
__device__ __forceinline__ uint32_t f32_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// rare path: a (distance,row) that passed the lane's distance gate
__device__ __forceinline__ LaneTop topk_insert(const uint64_t *__restrict__ labels, const uint64_t *__restrict__ allow_bits,
                                            uint64_t allow_nbits, uint32_t k, float dist, uint32_t row, float *list_d,
                                            uint64_t *list_l, uint32_t *qbound_q, LaneTop t) {
  const uint64_t lab = labels[row];
  if (!allow_bit(allow_bits, allow_nbits, lab)) return t;
  if (t.cnt >= k && !dl_less(dist, lab, t.thr_d, t.thr_l)) return t;
  const uint32_t at = t.cnt < k ? t.cnt++ : t.thr_i;
  list_d[at] = dist;
  list_l[at] = lab;
  if (t.cnt == k) {   // recompute this lane's worst entry
    float wd = list_d[0];
    uint64_t wl = list_l[0];
    uint32_t wi = 0;
    for (uint32_t i = 1; i < k; ++i) {
      const float di = list_d[i];
      const uint64_t lv = list_l[i];
      if (dl_less(wd, wl, di, lv)) { wd = di; wl = lv; wi = i; }
    }
    t.thr_d = wd;
    t.thr_l = wl;
    t.thr_i = wi;
    // k kept entries are all <= wd: nothing beyond wd can reach this query's final top-k, whichever
    // list it would land in -- publish it for every other lane working on the query
    __hip_atomic_fetch_min(qbound_q, f32_key(wd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return t;
}

// k <= kRegCap: the lane's list lives in registers (distance + row index; labels are looked up only to
// break distance ties, to apply the filter, and when the list is written out).  An insert is ~50
// VALU operations and no memory round trip; with the HBM-resident list every insert of any lane
// stalled the whole wave -- and its matrix pipe -- for about a microsecond, ~6000 times per wave at
// 10M rows.
constexpr int kRegCap = 10;
struct RegTop {
  float d[kRegCap];
  uint32_t r[kRegCap];
  float thr_d;            // worst kept distance, +inf while the list is not full
  uint32_t thr_r, thr_i, cnt;
};
__device__ __forceinline__ RegTop reg_insert(const uint64_t *__restrict__ labels, const uint64_t *__restrict__ allow_bits,
                                          uint64_t allow_nbits, uint32_t k, float dist, uint32_t row, RegTop s) {
  if (allow_bits != nullptr && !allow_bit(allow_bits, allow_nbits, labels[row])) return s;
  if (s.cnt >= k && dist == s.thr_d && !(labels[row] < labels[s.thr_r])) return s;   // tie: label decides
  const uint32_t at = s.cnt < k ? s.cnt++ : s.thr_i;
#pragma unroll
  for (int i = 0; i < kRegCap; ++i) {
    s.d[i] = (uint32_t)i == at ? dist : s.d[i];
    s.r[i] = (uint32_t)i == at ? row : s.r[i];
  }
  if (s.cnt == k) {   // recompute the worst entry: largest (distance, label)
    float wd = s.d[0];
    uint32_t wr = s.r[0], wi = 0;
#pragma unroll
    for (int i = 1; i < kRegCap; ++i) {
      if ((uint32_t)i < k) {
        const float di = s.d[i];
        bool worse = di > wd;
        if (di == wd) worse = labels[s.r[i]] > labels[wr];
        wd = worse ? di : wd;
        wr = worse ? s.r[i] : wr;
        wi = worse ? (uint32_t)i : wi;
      }
    }
    s.thr_d = wd;
    s.thr_r = wr;
    s.thr_i = wi;
  }
  return s;
}

// Position in this block's flattened (tile, stage) stream, kept incrementally (no divisions).
struct StreamPos {
  uint32_t tile_row0;   // first row of the tile
  uint32_t st;          // stage inside the tile
  uint32_t left;        // stages still to come, this one included
};
__device__ __forceinline__ void stream_advance(StreamPos &p, uint32_t stages, uint32_t tile_step_rows) {
  // branch-free (a stage of the main loop must stay one basic block for the instruction interleave).  The position
  // never moves past the LAST stage of the stream: the prefetches issued behind the end re-read that stage (it used
  // to step on to the following tile, up to 128 rows past the store's slack: a fault when the allocation ends on a
  // page boundary right there -- 4362 rows of 576 floats did)
  const bool go = p.left > 1;
  const bool wrap = go && p.st + 1 == stages;
  p.left -= p.left != 0 ? 1u : 0u;
  p.st = wrap ? 0u : p.st + (go ? 1u : 0u);
  p.tile_row0 += wrap ? tile_step_rows : 0u;
}

// Register-resident staging sets are plain structs handled BY VALUE (arrays passed by reference
// end up in scratch memory once the kernel is at its VGPR budget, and every load is then waited
// for immediately).
struct Stg { float4 v0, v1, v2, v3; };
struct Frag { float4 a0, a1, a2, a3, b0, b1, b2, b3; };

// global -> registers for one stage of ONE WAVE: its 32 rows x 2 chunks = 256 float4, 4 per lane:
// idx = lane + 64*u -> row idx/8 (of the wave's 32), 16-B column idx%8 of the 2-chunk slab.
// Address = wave-uniform base (tile row, stage: scalar registers) + a per-lane offset that never
// changes (LaneOff, computed once), so a stage issues its four loads without any vector address
// arithmetic.  No clamping: the row store keeps RowStore::kRowSlack readable rows past its capacity,
// rows past n_rows are masked in the epilogue.  p.tile_row0 already includes the wave's row offset.
struct LaneOff { uint32_t o0, o1, o2, o3; };   // in elements
// bf16 rows (kBf16): a stage of a row is 32 elements = 64 B, so the wave's 32 x 64 B are 128 16-B
// pieces, 2 per lane: idx = lane + 64*u -> row idx/4, piece idx%4 (8 elements each); they are widened
// to f32 (<< 16, exact) on the way into LDS, so everything behind the staging ring is the f32 kernel.
// which of the first 16 rows a lane stages (4 lanes per row): the 16 lanes that share an LDS store pass
// take rows {2G, 2G+1, 2G+8, 2G+9}, so their 16-B stores land in 16 different bank groups with the
// 36-dword row stride ((9*row + 2*piece) mod 16 is then a permutation) -- rows in plain order conflict 2-way
__device__ __forceinline__ uint32_t bf16_lane_row(uint32_t lane) {
  const uint32_t G = lane >> 4, rr = (lane >> 2) & 3;
  return 2 * G + (rr & 1) + 8 * (rr >> 1);
}
template <bool kBf16>
__device__ __forceinline__ LaneOff lane_offsets(uint32_t lane, uint32_t row_stride_e) {
  if constexpr (kBf16) {
    const uint32_t o = bf16_lane_row(lane) * row_stride_e + (lane & 3) * 8;
    return LaneOff{o, o + 16 * row_stride_e, 0, 0};
  } else {
    const uint32_t o = (lane >> 3) * row_stride_e + (lane & 7) * 4;
    return LaneOff{o, o + 8 * row_stride_e, o + 16 * row_stride_e, o + 24 * row_stride_e};
  }
}
template <bool kBf16>
__device__ __forceinline__ Stg stage_load(const FlatGemmArgs &a, const LaneOff lo, const StreamPos &p) {
  Stg s;
  if constexpr (kBf16) {
    const uint16_t *base = static_cast<const uint16_t *>(a.rows) + (size_t)p.tile_row0 * a.row_stride_f + p.st * 32;
    s.v0 = *reinterpret_cast<const float4 *>(base + lo.o0);
    s.v1 = *reinterpret_cast<const float4 *>(base + lo.o1);
    s.v2 = s.v0;
    s.v3 = s.v0;
  } else {
    // chunk = st*2 + c4/4, 16-B piece c4%4 of it == float offset st*32 + c4*4 (rows are zero padded
    // to whole stages, so there is no tail)
    const float *base = static_cast<const float *>(a.rows) + (size_t)p.tile_row0 * a.row_stride_f + p.st * 32;
    s.v0 = *reinterpret_cast<const float4 *>(base + lo.o0);
    s.v1 = *reinterpret_cast<const float4 *>(base + lo.o1);
    s.v2 = *reinterpret_cast<const float4 *>(base + lo.o2);
    s.v3 = *reinterpret_cast<const float4 *>(base + lo.o3);
  }
  return s;
}

// eight bf16 (one 16-B piece) -> two float4
__device__ __forceinline__ void widen8(const float4 raw, float4 &lo, float4 &hi) {
  const uint32_t w0 = __float_as_uint(raw.x), w1 = __float_as_uint(raw.y), w2 = __float_as_uint(raw.z),
                 w3 = __float_as_uint(raw.w);
  lo = float4{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
              __uint_as_float(w1 & 0xffff0000u)};
  hi = float4{__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xffff0000u), __uint_as_float(w3 << 16),
              __uint_as_float(w3 & 0xffff0000u)};
}

template <bool kBf16>
__device__ __forceinline__ void stage_store(float *buf, uint32_t lane, const Stg s) {
  if constexpr (kBf16) {
    // The raw pieces are made opaque HERE, one stage after their load was issued: otherwise the compiler widens a
    // piece right behind its global load and every stage waits (vmcnt(0)) for the HBM round trip it has just started.
    float4 r0 = s.v0, r1 = s.v1;
    asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
    float4 a0, a1, b0, b1;
    widen8(r0, a0, a1);
    widen8(r1, b0, b1);
    float *p0 = buf + bf16_lane_row(lane) * kXStride + (lane & 3) * 8;
    float *p1 = p0 + 16 * kXStride;
    *reinterpret_cast<float4 *>(p0) = a0;
    *reinterpret_cast<float4 *>(p0 + 4) = a1;
    *reinterpret_cast<float4 *>(p1) = b0;
    *reinterpret_cast<float4 *>(p1 + 4) = b1;
  } else {
    *reinterpret_cast<float4 *>(buf + ((lane) >> 3) * kXStride + ((lane) & 7) * 4) = s.v0;
    *reinterpret_cast<float4 *>(buf + ((lane + 64) >> 3) * kXStride + ((lane + 64) & 7) * 4) = s.v1;
    *reinterpret_cast<float4 *>(buf + ((lane + 128) >> 3) * kXStride + ((lane + 128) & 7) * 4) = s.v2;
    *reinterpret_cast<float4 *>(buf + ((lane + 192) >> 3) * kXStride + ((lane + 192) & 7) * 4) = s.v3;
  }
}

// LDS -> registers: this lane's A (row) and B (query) operands of one stage, 4 class groups each
__device__ __forceinline__ Frag frag_load(const float *xb, const float *q_row, uint32_t st, uint32_t kk) {
  const float *qb = q_row + (st * 2 + kk) * 16;
  Frag f;
  f.a0 = *reinterpret_cast<const float4 *>(xb);
  f.a1 = *reinterpret_cast<const float4 *>(xb + 4);
  f.a2 = *reinterpret_cast<const float4 *>(xb + 8);
  f.a3 = *reinterpret_cast<const float4 *>(xb + 12);
  f.b0 = *reinterpret_cast<const float4 *>(qb);
  f.b1 = *reinterpret_cast<const float4 *>(qb + 4);
  f.b2 = *reinterpret_cast<const float4 *>(qb + 8);
  f.b3 = *reinterpret_cast<const float4 *>(qb + 12);
  return f;
}

#define VK_MFMA4(P, AV, BV)                                                                                      \
  acc[4 * P + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, BV.x, kZeroC ? zero : acc[4 * P + 0], 0, 0, 0);  \
  acc[4 * P + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, BV.y, kZeroC ? zero : acc[4 * P + 1], 0, 0, 0);  \
  acc[4 * P + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, BV.z, kZeroC ? zero : acc[4 * P + 2], 0, 0, 0);  \
  acc[4 * P + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, BV.w, kZeroC ? zero : acc[4 * P + 3], 0, 0, 0);

template <bool kZeroC>
__device__ __forceinline__ void stage_mfma(f32x16 (&acc)[16], const Frag f) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  VK_MFMA4(0, f.a0, f.b0)
  VK_MFMA4(1, f.a1, f.b1)
  VK_MFMA4(2, f.a2, f.b2)
  VK_MFMA4(3, f.a3, f.b3)
}

// kAblate (timing experiments only, results invalid when != 0), cumulative: 1 = the top-k insert is
// replaced by a running minimum (no divergent path), 2 = and the HBM loads re-read one hot tile (L2
// hits only), 3 = no global loads at all, 4 = no LDS stores, 5 = no LDS fragment reads (MFMA +
// epilogue arithmetic only)
// kMode: 7 = the stage's memory operations pinned between the MFMAs (default), 0 = their placement left
// to the compiler.  (Tried and dropped: staging straight into LDS with global_load_lds_dwordx4 and a
// swizzled unpadded ring -- bit-exact, but a DMA completes well over two stages after issue (9 ms of
// vmcnt stalls per launch; LDS has no room for a deeper ring), and even with the wait removed it was
// only 1.8 ms faster than the register path.)
template <int kAblate, int kMode, bool kRegList, bool kBf16>
__device__ __forceinline__ void flat_gemm_body(const FlatGemmArgs &a) {
  extern __shared__ float lds[];
  if (launch_skipped(a.run_flag, a.run_if, a.run_hi)) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it in a scalar register
  const uint32_t li = lane & 31;   // row (A) / query (B) index inside the 32-wide MFMA tile
  const uint32_t kk = lane >> 5;   // which of the two chunks of the stage this lane feeds
  const uint32_t chunks = a.chunks;
  // stages of 2 chunks each; the row stride is a multiple of 4 chunks, so the count is even and the
  // ping-pong register sets keep their parity from tile to tile (padding chunks are zeros)
  const uint32_t stages = chunks / 2;
  const uint32_t qstride = a.row_stride_f + 4;

  float *lds_q = lds;                                   // [32][qstride], shared, read-only after the first barrier
  constexpr uint32_t kBufFloats = 32 * kXStride;        // one stage of one wave: 32 rows x 36 dwords
  // each wave stages ITS OWN 32 rows through a private ring of kXBufs buffers: no block barrier in
  // the main loop (a barrier idles the matrix pipe of a one-wave-per-SIMD kernel every stage)
  // queries per block: 32 (the MFMA tile's width) while the resident tile fits LDS next to the staging rings,
  // 24 or 16 for long rows (D = 1024, 1536, ...): the missing columns re-read the last resident query and their
  // results are discarded -- matrix-core efficiency drops with them, the arithmetic of the live columns does not
  const uint32_t tq = a.tile_q;
  float *lds_x = lds + (size_t)tq * qstride + (size_t)wave * kXBufs * kBufFloats;

  const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
  const uint32_t rp = (seq / a.nqt) * 8u + xcd;
  const uint32_t qt = seq % a.nqt;
  const uint32_t q0 = qt * tq;

  // ---- Q tile -> LDS (queries past nq replicate the last one; their results are discarded)
  for (uint32_t i = tid; i < tq * (a.row_stride_f / 4); i += 256) {
    const uint32_t q = i / (a.row_stride_f / 4), c4 = i % (a.row_stride_f / 4);
    const uint32_t gq = q0 + q < a.nq ? q0 + q : a.nq - 1;
    const float4 v = reinterpret_cast<const float4 *>(a.queries + (size_t)gq * a.q_stride_f)[c4];
    *reinterpret_cast<float4 *>(lds_q + (size_t)q * qstride + c4 * 4) = v;
  }

  // ---- this lane's private top-k list (HBM scratch) and threshold
  const uint32_t my_q = q0 + li;
  const bool q_valid = li < tq && my_q < a.nq;
  const uint32_t slot = wave * 2 + kk;
  const size_t list_base = (((size_t)(q_valid ? my_q : 0) * a.nrp + rp) * 8 + slot) * a.k;
  float *list_d = a.part_dist + list_base;
  uint64_t *list_l = a.part_label + list_base;
  if (q_valid && !kRegList)
    for (uint32_t i = 0; i < a.k; ++i) { list_d[i] = __builtin_inff(); list_l[i] = kNoLabel; }
  LaneTop top{__builtin_inff(), kNoLabel, 0u, 0u};
  RegTop rtop;
#pragma unroll
  for (int i = 0; i < kRegCap; ++i) { rtop.d[i] = __builtin_inff(); rtop.r[i] = 0xffffffffu; }
  // a pre-pass over the first rows bounds the k-th best distance: the list starts with that gate instead of
  // accepting everything until it has filled (3x fewer insert events)
  rtop.thr_d = a.init_bound ? a.init_bound[q_valid ? my_q : 0] : __builtin_inff();
  rtop.thr_r = 0xffffffffu;
  rtop.thr_i = 0;
  rtop.cnt = 0;
  uint32_t *qbound_q = a.qbound + (q_valid ? my_q : 0);

  // row tiles of this partition: a contiguous range (consecutive tiles are adjacent in memory: fewer
  // TLB fills and DRAM page changes than the strided assignment rp, rp+nrp, ...)
  const uint32_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
  const uint32_t t_base = n_tiles / a.nrp, t_rem = n_tiles % a.nrp;
  const uint32_t first_tile = a.contig ? rp * t_base + (rp < t_rem ? rp : t_rem) : rp;
  const uint32_t my_tiles = a.contig ? t_base + (rp < t_rem ? 1u : 0u) : (rp < n_tiles ? (n_tiles - rp + a.nrp - 1) / a.nrp : 0);
  const uint32_t total = my_tiles * stages;             // stages in this block's stream (< 2^32: <= 2^25 tiles)
  const uint32_t tile_step_rows = a.contig ? kTileRows : a.nrp * kTileRows;
  if (total == 0) {   // more partitions than tiles: this block only owes the merge its empty lists
    if (q_valid && kRegList)
      for (uint32_t i = 0; i < a.k; ++i) { list_d[i] = __builtin_inff(); list_l[i] = kNoLabel; }
    return;
  }

  // Software pipeline over the stream, iteration i = stage i:
  //   global loads for stage i+3 are issued at the top of i, written to LDS at the bottom of i+1
  //   (two iterations of cover), read LDS->registers during i+2, multiplied during i+3.
  // LDS ring: stage s lives in buffer s % 3; the buffer written at the bottom of i (stage i+2)
  // last held stage i-1, whose fragments were fetched during i-2.  One barrier per iteration.
  // The loop is unrolled by two with ping-pong register sets (no register rotation: a move of
  // a register that is the target of an in-flight load would wait for the load).
  const LaneOff loff = lane_offsets<kBf16>(lane, a.row_stride_f);
  StreamPos ld{first_tile * kTileRows + wave * 32, 0, total};   // next stage to fetch from HBM (this wave's rows)
  Stg stg_a, stg_b;
  // prologue: stages 0 and 1 straight to LDS, the next one left in registers.  A
  // stream has at least two stages; loads past its end re-read the last stage and are unused.
  stg_a = stage_load<kBf16>(a, loff, ld);
  stage_store<kBf16>(lds_x, lane, stg_a);
  stream_advance(ld, stages, tile_step_rows);
  stg_a = stage_load<kBf16>(a, loff, ld);
  stage_store<kBf16>(lds_x + kBufFloats, lane, stg_a);
  stream_advance(ld, stages, tile_step_rows);
  stg_a = stage_load<kBf16>(a, loff, ld);
  stg_b = stg_a;
  stream_advance(ld, stages, tile_step_rows);
  __syncthreads();                                       // the shared Q tile is in place

  const uint32_t x_off = li * kXStride + kk * 16;
  const float *q_row = lds_q + (size_t)(li < tq ? li : tq - 1) * qstride;
  Frag f0 = frag_load(lds_x + x_off, q_row, 0, kk), f1 = f0;

  uint32_t rbuf = 1;            // buffer of stage done+1
  uint32_t wbuf = 2;            // buffer of stage done+2
  uint32_t tile_row0 = first_tile * kTileRows;

  // one pipeline iteration: F = fragments of stage ST, NF receives the next stage's, SNEW receives
  // the loads of stage done+3, SOLD (loads of stage done+2) goes to LDS.  Past the end of the
  // stream the prefetches are harmless re-reads (ld stops advancing, results unused).
#define VK_GEMM_STAGE(ZERO, ST, F, NF, SNEW, SOLD)                                                \
  {                                                                                               \
    /* HBM loads for a later stage, LDS fragment reads for stage +1, this stage's 16 MFMAs        */ \
    /* (operands fetched one stage ago), LDS store of the oldest loads in flight.                 */ \
    const uint32_t nst = (ST) + 1 == stages ? 0u : (ST) + 1;                                      \
    if constexpr (kAblate < 2) SNEW = stage_load<kBf16>(a, loff, ld);                                    \
    if constexpr (kAblate == 2) { StreamPos hot = ld; hot.tile_row0 = wave * 32; SNEW = stage_load<kBf16>(a, loff, hot); } \
    stream_advance(ld, stages, tile_step_rows);                                                   \
    {                                                                                             \
      stage_mfma<ZERO>(acc, F);                                                                   \
      if constexpr (kAblate < 5) NF = frag_load(lds_x + rbuf * kBufFloats + x_off, q_row, nst, kk); \
      if constexpr (kAblate < 4) stage_store<kBf16>(lds_x + wbuf * kBufFloats, lane, SOLD);              \
    }                                                                                             \
    if constexpr (kMode == 7) {                                                            \
      /* M V M V M V M V | R R R R | M M M M | R R | M W M W M W M W | R R | M M M M: every       */ \
      /* fragment of the next stage is requested >= 4 MFMAs before this stage ends                 */ \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                        \
      }                                                                                           \
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                          \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
        if constexpr (kBf16) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   /* widen one float4 */ \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                        \
      }                                                                                           \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    rbuf = rbuf == 2 ? 0u : rbuf + 1;                                                             \
    wbuf = wbuf == 2 ? 0u : wbuf + 1;                                                             \
  }

  // Lockstep with the other query-tile blocks that stream the same rows (same rp, same wave index,
  // all on this XCD): each wave publishes the tile it has started and does not start tile t before
  // every sharer has started tile t-W.  Without it the sharers drift further apart than the XCD's
  // 4 MB L2 holds (4 panels x 393 KB per tile generation) and every one of them misses to HBM.
  // A performance hint only: after kMaxSpins polls the wave stops waiting for good (a grid that is
  // not fully co-resident must not hang).
  uint32_t lockstep = a.lockstep;
  uint32_t *sync_grp = a.sync + ((size_t)rp * 4 + wave) * 32;
  constexpr uint32_t kMaxSpins = 4096;

  for (uint32_t t = 0; t < my_tiles; ++t) {
    // cancellation word (host memory): requested at the start of every kCancelPollTiles-th tile, looked at behind its
    // epilogue -- the PCIe round trip hides behind the tile
    uint32_t cancel_now = 0;
    if (a.cancel && (t % kCancelPollTiles) == 0) cancel_now = __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lockstep && lane == 0) __hip_atomic_store(sync_grp + qt, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the query's shared pruning bound, fetched now and used a whole tile later in the epilogue
    uint32_t bkey = 0xFF800000u;
    if constexpr (!kRegList) bkey = __hip_atomic_load(qbound_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x16 acc[16];
    {
      VK_GEMM_STAGE(true, 0u, f0, f1, stg_b, stg_a)    // accumulators are born from a zero C operand
      VK_GEMM_STAGE(false, 1u, f1, f0, stg_a, stg_b)
      for (uint32_t st = 2; st < stages; st += 2) {
        VK_GEMM_STAGE(false, st, f0, f1, stg_b, stg_a)
        VK_GEMM_STAGE(false, st + 1, f1, f0, stg_a, stg_b)
      }
    }
    // progress of the sharers, fetched behind the epilogue and looked at after it
    uint32_t seen = 0xffffffffu;
    if (lockstep && lane < a.nqt) seen = __hip_atomic_load(sync_grp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float gate_b = key_f32(bkey);
    // ---- tile done.  Per output register r: gather the 16 class sums, combine them with
    // _mm512_reduce_add_ps's pairing (l,l+8) -> (l,l+4) -> (l,l+2) -> (0,1), 1 - dot, gate
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v[16];
#pragma unroll
      for (int l = 0; l < 16; ++l) v[l] = acc[l][r];
#pragma unroll
      for (int l = 0; l < 8; ++l) v[l] = v[l + 8] + v[l];
#pragma unroll
      for (int l = 0; l < 4; ++l) v[l] = v[l + 4] + v[l];
      const float dot = (v[0] + v[2]) + (v[1] + v[3]);
      const float dist = 1.0f - dot;
      const uint32_t row = tile_row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if constexpr (kAblate >= 1) {
        top.thr_d = dist < top.thr_d ? dist : top.thr_d;
        top.thr_i = row;
      } else if constexpr (kRegList) {
        if (q_valid && row < a.n_rows && dist <= rtop.thr_d)
          rtop = reg_insert(a.labels, a.allow_bits, a.allow_nbits, a.k, dist, row, rtop);
      } else if (q_valid && row < a.n_rows && dist <= top.thr_d && dist <= gate_b) {
        top = topk_insert(a.labels, a.allow_bits, a.allow_nbits, a.k, dist, row, list_d, list_l, qbound_q, top);
      }
      // keep the 16 gathers of one output register together: hoisting all 256 accumulator reads
      // ahead of the adds would need 256 VGPRs and spill into the pipelined loop
      __builtin_amdgcn_sched_barrier(0);
    }
    tile_row0 += tile_step_rows;
    if (cancel_now) break;   // (bruteforce.h:129) the lists keep what they hold
    if (lockstep && t + 1 < my_tiles) {
      // about to start tile t+1: everyone must have started tile t+1-W, i.e. published >= t+2-W
      const uint32_t need = t + 2 > lockstep ? t + 2 - lockstep : 0;
      uint32_t spins = 0;
      while (__builtin_amdgcn_ballot_w64(seen < need) != 0) {
        if (++spins > kMaxSpins) { lockstep = 0; break; }
        __builtin_amdgcn_s_sleep(4);
        if (lane < a.nqt) seen = __hip_atomic_load(sync_grp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if constexpr (kAblate >= 1) {   // keep the arithmetic alive
    if (q_valid) { list_d[0] = top.thr_d; list_l[0] = top.thr_i; }
  } else if constexpr (kRegList) {
    if (q_valid) {
#pragma unroll
      for (int i = 0; i < kRegCap; ++i)
        if ((uint32_t)i < a.k) {
          list_d[i] = rtop.d[i];
          list_l[i] = rtop.r[i] == 0xffffffffu ? kNoLabel : a.labels[rtop.r[i]];
        }
    }
  }
#undef VK_GEMM_STAGE
#undef VK_MFMA4
}

template <int kAblate, int kMode, bool kRegList, bool kBf16>
__global__ __launch_bounds__(256, 1) void flat_gemm_kernel(FlatGemmArgs a) {
  flat_gemm_body<kAblate, kMode, kRegList, kBf16>(a);
}
// the same code under its own name for the pre-pass over the first rows (FlatIndex::scan_gemm), so that
// per-kernel profiles do not average a 0.1 ms launch into the 37 ms one
template <bool kRegList, bool kBf16>
__global__ __launch_bounds__(256, 1) void flat_gemm_prepass_kernel(FlatGemmArgs a) {
  flat_gemm_body<0, 7, kRegList, kBf16>(a);
}

size_t flat_gemm_lds_bytes(uint32_t row_stride_f, uint32_t tile_q) {
  return ((size_t)tile_q * (row_stride_f + 4) + (size_t)kXBufs * kTileRows * kXStride) * 4;
}

// queries per block for this row length: 32, 24 or 16 (0 = the kernel does not fit)
uint32_t flat_gemm_tile_q(uint32_t row_stride_f) {
  for (uint32_t tq : {32u, 24u, 16u})
    if (flat_gemm_lds_bytes(row_stride_f, tq) <= 160 * 1024) return tq;
  return 0;
}

bool flat_gemm_supported(uint32_t row_stride_f, uint64_t k) {
  return (row_stride_f % 64) == 0 && flat_gemm_tile_q(row_stride_f) != 0 && k >= 1 && k <= 256;   // k > 10: per-lane lists in HBM scratch (an insert costs O(k): beyond 256 the scan wins)
}

hipError_t launch_flat_gemm(const FlatGemmArgs &a, hipStream_t s) {
  if (a.nrp == 0 || (a.nrp & 7u) || a.tile_q == 0 || a.nqt != (a.nq + a.tile_q - 1) / a.tile_q) return hipErrorInvalidValue;
  const size_t lds = flat_gemm_lds_bytes(a.row_stride_f, a.tile_q);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
#ifdef VK_EXPERIMENTS
  // the experiments build only (scripts/k4_ablate.sh): the timing ladder -- results INVALID -- and the A/B variants with
  // the compiler's own placement of a stage's memory operations / without the register lists
  static const int ablate = getenv("VK_GEMM_ABLATE") ? atoi(getenv("VK_GEMM_ABLATE")) : 0;
  static const int mode_env = getenv("VK_GEMM_MODE") ? atoi(getenv("VK_GEMM_MODE")) : 7;
  const int mode = mode_env == 7 ? 7 : 0;
  static const int reg_env = getenv("VK_GEMM_REGLIST") ? atoi(getenv("VK_GEMM_REGLIST")) : 1;
#else
  constexpr int ablate = 0, mode = 7, reg_env = 1;
#endif
  const bool reg = reg_env && a.k <= (uint32_t)kRegCap;
  const void *fn = a.prepass && !ablate ? (a.bf16 ? (reg ? reinterpret_cast<const void *>(&flat_gemm_prepass_kernel<true, true>)
                                                          : reinterpret_cast<const void *>(&flat_gemm_prepass_kernel<false, true>))
                                                  : (reg ? reinterpret_cast<const void *>(&flat_gemm_prepass_kernel<true, false>)
                                                          : reinterpret_cast<const void *>(&flat_gemm_prepass_kernel<false, false>)))
#ifdef VK_EXPERIMENTS
                 : ablate == 1 ? reinterpret_cast<const void *>(&flat_gemm_kernel<1, 7, false, false>)
                 : ablate == 2 ? reinterpret_cast<const void *>(&flat_gemm_kernel<2, 7, false, false>)
                 : ablate == 3 ? reinterpret_cast<const void *>(&flat_gemm_kernel<3, 7, false, false>)
                 : ablate == 4 ? reinterpret_cast<const void *>(&flat_gemm_kernel<4, 7, false, false>)
                 : ablate == 5 ? reinterpret_cast<const void *>(&flat_gemm_kernel<5, 7, false, false>)
                 : mode != 7 && !a.bf16 ? (reg ? reinterpret_cast<const void *>(&flat_gemm_kernel<0, 0, true, false>)
                                               : reinterpret_cast<const void *>(&flat_gemm_kernel<0, 0, false, false>))
#endif
                 : a.bf16      ? (reg ? reinterpret_cast<const void *>(&flat_gemm_kernel<0, 7, true, true>)
                                      : reinterpret_cast<const void *>(&flat_gemm_kernel<0, 7, false, true>))
                               : (reg ? reinterpret_cast<const void *>(&flat_gemm_kernel<0, 7, true, false>)
                                      : reinterpret_cast<const void *>(&flat_gemm_kernel<0, 7, false, false>));
  (void)mode;
  hipError_t e = ensure_max_lds(fn);
  if (e != hipSuccess) return e;
  FlatGemmArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(fn, dim3(a.nrp * a.nqt), dim3(256), params, lds, s);
}

}  // namespace vk
