// flat_scan.hip -- K3: the FLAT brute-force scan (hnswlib::BruteforceSearch::searchKnn,
// third_party/hnswlib/bruteforce.h:116-145) as an HBM-streaming CDNA4 kernel, plus the
// per-query merge of the per-wave partial top-k lists and the label gather-scan used by
// the pre-filter path (src/indexes/vector_base.cc:509-530) and by single-record distance
// (vector_flat.cc:256-271).
//
// Roofline: memory bound.  Algorithmic bytes per pass = n_rows * row_stride (+ n/8 when
// a filter bitmap is consulted); kQB queries share one pass.
//
// Mapping (see device_common.hpp for why): a wave owns 16 rows at a time, one per quad;
// lane j of a quad streams the row with 16-B loads at chunk*64 + j*16, so every load
// instruction of the wave touches 16 rows x 64 contiguous bytes and the whole row is
// consumed exactly once.  The query block lives in LDS and is read with quad-broadcast
// ds_read_b128.  Distances come out bit-identical to the reference CPU path, ties are
// resolved by (distance,label) exactly like the std::pair heap of bruteforce.h.
#include <algorithm>

#include <stdlib.h>

#include <type_traits>

#include "device_common.hpp"
#include "kernels.hpp"

#ifndef VK_SCAN_XB
#define VK_SCAN_XB(QB, L2) ((QB) >= 8 ? 4 : 8)
#endif

namespace vk {

// kLb: the scan carries an exclusive lower bound (distance,label) per query (FlatIndex::search_in_passes, k > 1024:
// always kQB == 1, kE == 16); without it the hot loop and the register budget do not pay for the paging feature
// kIdx: re-rank mode (FlatScanArgs::cand_row): the rows of the block's one query come from its survivor list
template <int kQB, bool kL2, int kE, bool kBf16, bool kLb = false, bool kIdx = false>
__global__ __launch_bounds__(256) void flat_scan_kernel(FlatScanArgs a) {
  extern __shared__ float4 qs[];  // [kQB][chunks][4] float4 == kQB padded queries
  if (launch_skipped(a.run_flag, a.run_if, a.run_hi)) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 3;
  const int rq = lane >> 2;
  const uint32_t chunks = a.chunks;
  // XCD-aware decode of the 1-D grid.  Hardware places consecutive block ids on
  // consecutive XCDs (id % 8); the blocks that re-read the SAME rows for different
  // query groups are made consecutive on ONE XCD so the re-reads hit its private L2:
  //   xcd = id % 8, s = id / 8, row partition = (s / nqg) * 8 + xcd, query group = s % nqg
  const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
  const uint32_t rp = (seq / a.nqg) * 8u + xcd;
  const uint32_t qbase = (seq % a.nqg) * kQB;
  // redo mode: the launch serves the *nq_dev queries listed in q_index (compact numbering from here on)
  uint32_t nq = a.nq;
  if constexpr (!kLb && !kIdx) {
    if (a.nq_dev) {
      const uint32_t n = *a.nq_dev;
      nq = n < nq ? n : nq;
      if (qbase >= nq) return;
    }
  }

  // stage the query block: queries past nq replicate the last one (results discarded)
  {
    const uint32_t per_q = chunks * 4;
    for (uint32_t i = threadIdx.x; i < per_q * kQB; i += blockDim.x) {
      uint32_t qi = i / per_q, off = i - qi * per_q;
      uint32_t q = qbase + qi < nq ? qbase + qi : nq - 1;
      if constexpr (!kLb && !kIdx) {
        if (a.q_index) q = a.q_index[q];
      }
      qs[i] = reinterpret_cast<const float4 *>(a.queries + (size_t)q * a.q_stride_f)[off];
    }
  }
  __syncthreads();

  WaveTopK<kE> top[kQB];
  float lbd[kQB];
  uint64_t lbl[kQB];
#pragma unroll
  for (int qi = 0; qi < kQB; ++qi) {
    top[qi].init(a.k);
    const uint32_t q = qbase + qi < nq ? qbase + qi : nq - 1;
    lbd[qi] = kLb ? a.lb_dist[q] : -__builtin_inff();
    lbl[qi] = kLb ? a.lb_label[q] : 0;
  }

  const uint32_t total_waves = a.nrp * 4;   // nrp is a multiple of 8
  const uint32_t wave_gid = rp * 4 + wave;
  uint32_t n_rows = a.row_end - a.row_begin;
  const uint32_t *cand = nullptr, *cand_chunks = nullptr;
  if constexpr (kIdx) {
    const uint32_t q = qbase < nq ? qbase : nq - 1;
    const uint32_t c = a.cand_cnt[q];
    const uint32_t most = a.cand_cap + (a.cand_qchunk ? kSpillPerQuery * kSpillChunk : 0u);
    n_rows = c < most ? c : most;
    if (a.cand_ovf && a.cand_ovf[q]) n_rows = 0;   // survivors were lost: the exact redo pass answers this query
    cand = a.cand_row + (size_t)q * a.cand_cap;
    cand_chunks = a.cand_qchunk ? a.cand_qchunk + (size_t)q * kSpillPerQuery : nullptr;
  }
  // entry i of the block's survivor list: the private list, then the query's spill chunks
  auto cand_at = [&](uint32_t i) -> uint32_t {
    if (i < a.cand_cap) return cand[i];
    const uint32_t jj = i - a.cand_cap;
    return a.cand_spill[(size_t)(cand_chunks[jj / kSpillChunk] - 2u) * kSpillChunk + jj % kSpillChunk];   // (slot = chunk + 2)
  };
  const uint32_t n_tiles = (n_rows + kRowsPerWave - 1) / kRowsPerWave;

  uint32_t polled = 0;
  for (uint32_t tile = wave_gid; tile < n_tiles; tile += total_waves) {
    if (a.cancel && (polled++ % kCancelPollTiles) == 0 && poll_cancel(a.cancel)) break;   // bruteforce.h:129
    uint32_t row, lrow;
    bool valid;
    if constexpr (kIdx) {
      const uint32_t i = tile * kRowsPerWave + rq;
      valid = i < n_rows;
      row = lrow = cand_at(valid ? i : n_rows - 1);
    } else {
      row = a.row_begin + tile * kRowsPerWave + rq;
      valid = row < a.row_end;
      lrow = valid ? row : a.row_end - 1;
    }
    const char *__restrict__ base = row_base<kBf16>(a.rows, lrow, a.row_stride_f);

    float4 acc[kQB];
#pragma unroll
    for (int qi = 0; qi < kQB; ++qi) acc[qi] = make_float4(0.f, 0.f, 0.f, 0.f);

    // row pieces in flight per lane: 8, or 4 where the accumulators of 8 queries need the registers
    // (list mode: a wave has one round of 16 survivors, nothing to overlap its memory round trips with -- three times
    // the pieces in flight, a third of the round trips)
    constexpr int kXB = VK_SCAN_XB(kQB, kL2);
    uint32_t c = 0;
    if constexpr (kIdx && kQB == 1) {
      constexpr int kXL = 24;
      for (; c + kXL <= chunks; c += kXL) {
        float4 x[kXL];
#pragma unroll
        for (int u = 0; u < kXL; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
        for (int u = 0; u < kXL; ++u) chunk_fma<kL2>(acc[0], x[u], qs[(c + u) * 4 + j]);
      }
    }
    for (; c + kXB <= chunks; c += kXB) {
      float4 x[kXB];
#pragma unroll
      for (int u = 0; u < kXB; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
      for (int u = 0; u < kXB; ++u) {
#pragma unroll
        for (int qi = 0; qi < kQB; ++qi) chunk_fma<kL2>(acc[qi], x[u], qs[(qi * chunks + c + u) * 4 + j]);
      }
    }
    for (; c < chunks; ++c) {
      float4 x = row_piece<kBf16>(base, c * 4 + j);
#pragma unroll
      for (int qi = 0; qi < kQB; ++qi) chunk_fma<kL2>(acc[qi], x, qs[(qi * chunks + c) * 4 + j]);
    }

#pragma unroll
    for (int qi = 0; qi < kQB; ++qi) {
      const float dist = finish_distance<kL2>(quad_reduce16(acc[qi]));
      // distance gate first, filter second -- the order of bruteforce.h:131-135
      const bool cand = valid && j == 0 && dist <= top[qi].thr_d && (!kLb || dist >= lbd[qi]);
      uint64_t mask = __ballot(cand);
      while (mask) {
        const int b = __ffsll((unsigned long long)mask) - 1;
        mask &= mask - 1;
        const float cd = readlane_f32(dist, b);
        if (!(cd <= top[qi].thr_d)) continue;
        const uint32_t crow = __builtin_amdgcn_readlane((int)row, b);
        const uint64_t cl = a.labels[crow];
        if (kLb && !dl_less(lbd[qi], lbl[qi], cd, cl)) continue;   // not beyond the previous pass
        if (!allow_bit(a.allow_bits, a.allow_nbits, cl)) continue;
        top[qi].insert(cd, cl, lane);
      }
    }
  }

  if constexpr (kE == 1) {
    // Block-level merge (k <= 64): waves 1..3 hand their lists to wave 0 through LDS, so the
    // grid leaves one partial list per BLOCK and the final merge kernel reads 4x fewer entries.
    __syncthreads();                                   // everyone is done with the query block in LDS
    float *md = reinterpret_cast<float *>(qs);         // reuse it: [kQB][3][k] distances, then labels
    uint64_t *ml = reinterpret_cast<uint64_t *>(md + ((kQB * 3 * a.k + 1) & ~1u));
    if (wave > 0) {
#pragma unroll
      for (int qi = 0; qi < kQB; ++qi)
        if ((uint32_t)lane < a.k) {
          md[(qi * 3 + wave - 1) * a.k + lane] = top[qi].d[0];
          ml[(qi * 3 + wave - 1) * a.k + lane] = top[qi].lab[0];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int qi = 0; qi < kQB; ++qi) {
        for (int w = 0; w < 3; ++w) {
          float dist = __builtin_inff();
          uint64_t lab = kNoLabel;
          if ((uint32_t)lane < a.k) { dist = md[(qi * 3 + w) * a.k + lane]; lab = ml[(qi * 3 + w) * a.k + lane]; }
          uint64_t mask = __ballot(lab != kNoLabel && dist <= top[qi].thr_d);
          while (mask) {
            const int b = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const float cd = readlane_f32(dist, b);
            if (!(cd <= top[qi].thr_d)) continue;
            top[qi].insert(cd, readlane_u64(lab, b), lane);
          }
        }
        const uint32_t q = qbase + qi;
        if (q < nq) {
          const size_t base = ((size_t)q * a.nrp + rp) * a.k;
          top[qi].store(a.part_dist + base, a.part_label + base, lane);
        }
      }
    }
  } else {
#pragma unroll
    for (int qi = 0; qi < kQB; ++qi) {
      const uint32_t q = qbase + qi;
      if (q < nq) {
        const size_t base = ((size_t)q * total_waves + wave_gid) * a.k;
        top[qi].store(a.part_dist + base, a.part_label + base, lane);
      }
    }
  }
}

// ---- fused re-rank + selection of the candidate filter's survivors (k <= 64) ------------------------------------------------
// One launch turns a query's survivor list (private list, then spill chunks; row slots with their approximate scores) into
// its ANSWER:
//   1. the second bound: from the survivors' own scores a lower bound of the k-th best exact score, and with it the few
//      entries of the list that can still be among the k best (about k + 1 of a few hundred);
//   2. exact distances of those -- the quad kernel's arithmetic, sixteen rows per wave and round, labels requested with the
//      rows (r03 looked a label up inside the serial insert loop: a dependent 8-byte read per kept row) --, every wave
//      keeping its k best in registers;
//   3. the block's four lists merged by its wave 0 through LDS, rank-sorted by (distance, label), written out.
// A short list (up to kRerankSoloMax survivors: the usual case) is served by ONE block per query -- part 0 --, the whole list
// in registers; the other kRerankParts - 1 blocks of the query leave at once.  A long list (a query on tens of thousands of
// duplicates of one vector: spill chunks) is walked by all kRerankParts blocks, a contiguous slice per wave, each block
// writes a partial list, and the block that finishes LAST for its query (a counter per query) merges the partial lists and
// writes the answer.  r03 ran steps 2 and 3 as two launches (flat_scan_kernel in list mode, merge_select_kernel).
// (profiles/r04_step_trace_*.log, profiles/r04_rerank_stamps.log: 143 us -> 32 us at 10M x 768.)
constexpr int kRerankParts = 8;
constexpr uint32_t kRerankSlots = 576;       // LDS row slots per wave: eight steps of 64 entries (a short list's share) + one more
constexpr uint32_t kRerankSoloMax = 2048;    // survivors one block re-ranks alone
template <bool kL2, bool kBf16>
__global__ __launch_bounds__(256) void flat_rerank_kernel(FlatScanArgs a, MergeArgs m) {
  extern __shared__ float4 qs[];  // the query ([chunks][4] float4), later the waves' lists
#ifdef VK_EXPERIMENTS
#define VK_STAMP(i) do { if (a.stamps && blockIdx.x < a.nq && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define VK_STAMP(i) do { } while (0)
#endif
  VK_STAMP(0);
  // Grid: the FIRST nq blocks are part 0 of the queries, the other 7 nq are parts 1..7 (query-major).  A short list is
  // served by its part 0 alone, so the blocks that do the work are dispatched first, all at once, and -- workgroups being
  // dealt to the eight XCDs round robin by their number -- evenly over the dies; the rest leave as they come.  (With
  // part = number % 8 every working block sat on XCD 0: the blocks of a batch started over 100 us, each running 28.)
  const uint32_t q = blockIdx.x < a.nq ? blockIdx.x : (blockIdx.x - a.nq) / (kRerankParts - 1);
  const uint32_t part = blockIdx.x < a.nq ? 0u : 1u + (blockIdx.x - a.nq) % (kRerankParts - 1);
  // a query the filter handed over (it lost survivors, or cannot go through f16): listed for the exact redo pass, which
  // owns its output
  if (m.ovf_q && m.ovf_q[q]) {
    if (part == 0 && threadIdx.x == 0) m.redo_list[atomicAdd(m.redo_cnt, 1u)] = q;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 3;
  const int rq = lane >> 2;
  const uint32_t chunks = a.chunks;
  const uint32_t c_raw = a.cand_cnt[q];
  const uint32_t most = a.cand_cap + (a.cand_qchunk ? kSpillPerQuery * kSpillChunk : 0u);
  const uint32_t n_rows = c_raw < most ? c_raw : most;
  VK_STAMP(1);
  const uint32_t *cand = a.cand_row + (size_t)q * a.cand_cap;
  const uint32_t *cand_chunks = a.cand_qchunk ? a.cand_qchunk + (size_t)q * kSpillPerQuery : nullptr;
  auto cand_at = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t {
    if (i < a.cand_cap) return cand[i];
    const uint32_t jj = i - a.cand_cap;
    return a.cand_spill[(size_t)(cand_chunks[jj / kSpillChunk] - 2u) * kSpillChunk + jj % kSpillChunk];   // (slot = chunk + 2)
  };
  auto val_at = [&](uint32_t i) __attribute__((always_inline)) -> float {
    if (i < a.cand_cap) return a.cand_val[(size_t)q * a.cand_cap + i];
    const uint32_t jj = i - a.cand_cap;
    return a.cand_spill_val[(size_t)(cand_chunks[jj / kSpillChunk] - 2u) * kSpillChunk + jj % kSpillChunk];
  };
  constexpr uint32_t kStep = kRerankParts * 4;
  const uint32_t first = part * 4u + (uint32_t)wave;
  uint32_t *my_rows = reinterpret_cast<uint32_t *>(qs + chunks * 4) + wave * kRerankSlots;   // [4 waves][kRerankSlots] behind the query
  const bool solo = n_rows <= kRerankSoloMax;   // a short list: the query's first block serves it alone
  if (solo && part != 0) return;
  for (uint32_t i = threadIdx.x; i < chunks * 4; i += blockDim.x)
    qs[i] = reinterpret_cast<const float4 *>(a.queries + (size_t)q * a.q_stride_f)[i];

  // ---- the second bound (every wave on its own: no barrier, no other wave's result) ----------------------------------
  // lo_i = score_i - margin_i is a lower bound of survivor i's exact score, hi_i = score_i + margin_i an upper bound
  // (margin = the query's error polynomial at the row's tile norm: what the gate of flat_filter.hip charges).  k distinct
  // rows reach the k-th largest lo, so it bounds the k-th best exact score from below; a survivor whose hi stays under it
  // cannot be among the k best (ties at the k-th score have hi >= score = bound: kept, and settled by the exact
  // (distance, label) order).  The bound comes from the first kPer * 64 survivors (any k distinct rows will do).
  const bool prune = a.cand_val != nullptr;
  float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
  float thr2 = -__builtin_inff();
  // the first kPer * 64 survivors of the list, held in registers: entry u * 64 + lane = (row slot, score, margin).  All of
  // a step's loads are requested before any is waited for (eight entries per lane at a time: list and scores in one round
  // trip, the tile norms in a second) -- taken one entry at a time this part alone was 90 us of the kernel.
  constexpr int kPer = 32;
  uint32_t rowv[kPer];
  float valv[kPer], marv[kPer];
  uint32_t n_sel = n_rows < (uint32_t)(kPer * kWave) ? n_rows : (uint32_t)(kPer * kWave);
  if (n_sel > a.cand_cap) n_sel = a.cand_cap;                // (the private list: no spill chunks in this part)
  const uint32_t nu = (n_sel + kWave - 1) / kWave;
  if (prune) co = a.qcoef[q];
  // (clamped indices instead of predicates or branches: entries past the list re-read its last one -- one address for the
  //  whole wave -- and every load of the step goes out back to back)
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const uint32_t i = (uint32_t)u * kWave + lane;
    const uint32_t ic = i < n_sel ? i : (n_sel ? n_sel - 1 : 0u);
    // (an EMPTY list -- every row of a query filtered out -- has nothing at entry 0 but what an earlier batch, or nobody, left
    //  there: not to be used as a row slot for the tile-norm look-up below)
    rowv[u] = n_sel ? cand[ic] : 0u;
    valv[u] = prune ? a.cand_val[(size_t)q * a.cand_cap + ic] : 0.f;
  }
#ifdef VK_EXPERIMENTS
  if (a.stamps && rowv[0] + rowv[kPer - 1] + __float_as_uint(valv[0]) == 0xFFFFFFF1u) return;   // (nothing: keeps the stamp behind the loads)
#endif
  VK_STAMP(2);
#pragma unroll
  for (int u = 0; u < kPer; ++u)
    marv[u] = prune ? filter_margin<kL2>(co.x, co.y, co.z, __uint_as_float(a.tile_norm[rowv[u] >> 7])) : 0.f;
#ifdef VK_EXPERIMENTS
  if (a.stamps && marv[0] + marv[kPer - 1] == 123.25f) return;
#endif
  VK_STAMP(3);
  if (prune && n_sel >= a.k) {
    uint32_t key[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const float lo = valv[u] - marv[u];
      // (0 = below every float: never counted.  NaN, +-inf: a tile outside f16 -- no information)
      key[u] = ((uint32_t)u * kWave + lane < n_sel && fabsf(lo) < __builtin_inff()) ? desc_key(lo) : 0u;
    }
    // T = the largest threshold with kBoundBits significant bits that k keys reach: a lower bound of the k-th largest key,
    // short of it by less than 2^-(kBoundBits - 9) relative -- a bound need not be tight to the last bit, and every bit is a
    // dependent step.  The eight compares of a group write eight scalar masks before any is counted (as `c += popc(ballot)`
    // each compare went through VCC and waited for the count before it: 530 cycles a bit).
    constexpr int kBoundBits = 20;
    uint32_t T = 0;
    for (int bit = 31; bit >= 32 - kBoundBits; --bit) {
      const uint32_t cnd = T | (1u << bit);
      uint32_t c = 0;
      auto count8 = [&](auto G) __attribute__((always_inline)) {
        constexpr int g = decltype(G)::value;
        unsigned long long mk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(mk[u]) : "s"(cnd), "v"(key[g + u]));
#pragma unroll
        for (int u = 0; u < 8; ++u) c += (uint32_t)__builtin_popcountll(mk[u]);
      };
      count8(std::integral_constant<int, 0>{});
      if (nu > 8) count8(std::integral_constant<int, 8>{});
      if (nu > 16) count8(std::integral_constant<int, 16>{});
      if (nu > 24) count8(std::integral_constant<int, 24>{});
      if (c >= a.k) T = cnd;
    }
    if (T != 0) {
      const float b = desc_key_float(T);
      // (2^-21 max(1, |b|): the rounding of the additions and subtractions on both sides, as in gate_thr)
      if (b == b) thr2 = b - 0x1p-21f * fmaxf(1.f, fabsf(b));
    }
  }
  VK_STAMP(4);
  __syncthreads();   // (the query is in LDS)
  VK_STAMP(5);

  WaveTopK<1> top;
  top.init(a.k);
  // one round: exact distances of the wave's (up to 16) rows in my_rows, one per quad
  auto round16 = [&](uint32_t off, uint32_t cnt) __attribute__((always_inline)) {
    const uint32_t row = my_rows[off + ((uint32_t)rq < cnt ? rq : 0)];
    const bool valid = (uint32_t)rq < cnt;
    const char *__restrict__ base = row_base<kBf16>(a.rows, row, a.row_stride_f);
    const uint64_t row_label = a.labels[row];        // (with the row, not behind its distance)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t c = 0;
    constexpr int kXL = 48;     // row pieces in flight per lane: a 768-element row is ONE round trip (the block is alone on its CU's registers)
    for (; c + kXL <= chunks; c += kXL) {
      float4 x[kXL];
#pragma unroll
      for (int u = 0; u < kXL; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
      for (int u = 0; u < kXL; ++u) chunk_fma<kL2>(acc, x[u], qs[(c + u) * 4 + j]);
    }
    for (; c + 8 <= chunks; c += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = row_piece<kBf16>(base, (c + u) * 4 + j);
#pragma unroll
      for (int u = 0; u < 8; ++u) chunk_fma<kL2>(acc, x[u], qs[(c + u) * 4 + j]);
    }
    for (; c < chunks; ++c) chunk_fma<kL2>(acc, row_piece<kBf16>(base, c * 4 + j), qs[c * 4 + j]);
    const float dist = finish_distance<kL2>(quad_reduce16(acc));
    // distance gate first, filter second -- the order of bruteforce.h:131-135
    const bool allowed = allow_bit(a.allow_bits, a.allow_nbits, row_label);
    uint64_t mask = __ballot(valid && j == 0 && dist <= top.thr_d && allowed);
    while (mask) {
      const int b = __ffsll((unsigned long long)mask) - 1;
      mask &= mask - 1;
      const float cd = readlane_f32(dist, b);
      if (!(cd <= top.thr_d)) continue;
      top.insert(cd, readlane_u64(row_label, b), lane);
    }
  };
  // The entries that pass the second bound wait in the wave's LDS slots for an exact distance, sixteen at a time.  A short
  // list (the usual case: a few hundred survivors, a dozen of them kept) is served by the query's FIRST block alone -- the
  // other seven leave at once and nothing crosses a block.
  uint32_t kept = 0, pend = 0, polled = 0;
  // a step's kept entries join the waiting ones in the wave's LDS slots ...
  auto take = [&](bool keep, uint32_t row) __attribute__((always_inline)) {
    const uint64_t km = __ballot(keep);
    if (keep) my_rows[pend + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = row;
    pend += (uint32_t)__popcll(km);
    kept += (uint32_t)__popcll(km);
  };
  // ... and every full sixteen of them gets its exact distances; what is left (< 16 entries) moves to the front
  auto flush = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();
    uint32_t o = 0;
    for (; o + kRowsPerWave <= pend; o += kRowsPerWave) round16(o, kRowsPerWave);
    if (o != 0) {
      const uint32_t left = pend - o;
      uint32_t v = 0;
      if ((uint32_t)lane < left) v = my_rows[o + lane];
      __builtin_amdgcn_wave_barrier();
      if ((uint32_t)lane < left) my_rows[lane] = v;
      __builtin_amdgcn_wave_barrier();
      pend = left;
    }
  };
  if (solo && n_rows <= n_sel) {
    // the whole list is in registers: wave w of the block takes the steps u = w, w + 4, ... (at most eight: 512 slots)
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      // (every wave looks at every step and keeps its own: a compile-time register index, a run-time mask)
      const bool own = (uint32_t)(u & 3) == (uint32_t)wave && (uint32_t)u * kWave + lane < n_rows;
      if ((uint32_t)u < nu) take(own && !(valv[u] + marv[u] < thr2), rowv[u]);   // ("not below": a NaN stays)
    }
  } else {
    // a long list (spill chunks, or a private list cut short by the option filter-cap): every wave of the query's blocks
    // walks its own contiguous slice of it, 64 entries per step
    const uint32_t n_waves = solo ? 4u : kStep, wid = solo ? (uint32_t)wave : first;
    const uint32_t per = (n_rows + n_waves - 1) / n_waves;
    const uint32_t s_lo = wid * per < n_rows ? wid * per : n_rows, s_hi = s_lo + per < n_rows ? s_lo + per : n_rows;
    for (uint32_t base = s_lo; base < s_hi; base += kWave) {
      if (a.cancel && (polled++ % kCancelPollTiles) == 0 && poll_cancel(a.cancel)) break;   // bruteforce.h:129
      const uint32_t i = base + lane;
      bool keep = i < s_hi;
      uint32_t row = 0;
      if (keep) {
        row = cand_at(i);
        if (prune) {
          const float R = __uint_as_float(a.tile_norm[row >> 7]);
          const float hi = val_at(i) + filter_margin<kL2>(co.x, co.y, co.z, R);
          keep = !(hi < thr2);
        }
      }
      take(keep, row);
      if (pend + kWave > kRerankSlots) flush();
    }
  }
  VK_STAMP(6);
  flush();
  VK_STAMP(7);
  if (pend != 0) round16(0, pend);
  if (a.reranked && lane == 0 && kept != 0) atomicAdd(a.reranked + q, kept);   // (per query: at most 32 waves meet here)

  // waves 1..3 -> LDS -> wave 0 (the query block is done with)
  __syncthreads();
  float *md = reinterpret_cast<float *>(qs);                               // [3][k] distances, then labels
  uint64_t *ml = reinterpret_cast<uint64_t *>(md + ((3 * a.k + 1) & ~1u));
  if (wave > 0 && (uint32_t)lane < a.k) {
    md[(wave - 1) * a.k + lane] = top.d[0];
    ml[(wave - 1) * a.k + lane] = top.lab[0];
  }
  __syncthreads();
  if (wave > 0) return;
  VK_STAMP(8);
  auto absorb = [&](float dist, uint64_t lab) __attribute__((always_inline)) {
    uint64_t mask = __ballot(lab != kNoLabel && dist <= top.thr_d);
    while (mask) {
      const int b = __ffsll((unsigned long long)mask) - 1;
      mask &= mask - 1;
      const float cd = readlane_f32(dist, b);
      if (!(cd <= top.thr_d)) continue;
      top.insert(cd, readlane_u64(lab, b), lane);
    }
  };
  for (int w = 0; w < 3; ++w) {
    float dist = __builtin_inff();
    uint64_t lab = kNoLabel;
    if ((uint32_t)lane < a.k) { dist = md[w * a.k + lane]; lab = ml[w * a.k + lane]; }
    absorb(dist, lab);
  }
  if (!solo) {
  // the block's partial list; the block that arrives last at the query's counter merges all of them
  float *pd = a.part_dist + ((size_t)q * kRerankParts + part) * a.k;
  uint64_t *pl = a.part_label + ((size_t)q * kRerankParts + part) * a.k;
  if ((uint32_t)lane < a.k) {
    __hip_atomic_store(reinterpret_cast<uint32_t *>(pd) + lane, __float_as_uint(top.d[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pl + lane, top.lab[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // Publication of the partial list: device-coherent stores, then a RELEASE bump of the query's counter at agent scope; the
  // block that arrives last ACQUIRES before it reads the other blocks' lists.  (r04 had no fences here -- it relied on the
  // sc1 stores passing the die's L2 and on the wave's s_waitcnt, formally a data race -- because an agent-scope release /
  // acquire writes back / invalidates a whole L2 and cost 134 -> 329 us when EVERY query merged across blocks.  Since the
  // second bound only lists over 2048 survivors come here; the fences are paid by those alone.)
  uint32_t arrived = 0;
  if (lane == 0) arrived = __hip_atomic_fetch_add(a.done_cnt + q, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
  if (arrived != kRerankParts - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (uint32_t p2 = 0; p2 < kRerankParts; ++p2) {
    if (p2 == part) continue;
    float dist = __builtin_inff();
    uint64_t lab = kNoLabel;
    if ((uint32_t)lane < a.k) {
      const size_t at = ((size_t)q * kRerankParts + p2) * a.k + lane;
      dist = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(a.part_dist) + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      lab = __hip_atomic_load(a.part_label + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    absorb(dist, lab);
  }
  }
  // rank sort of the kept entries by (distance, label) (labels are unique, so are the keys), padding, count
  const uint32_t cnt = top.cnt;
  float *od = m.out_dist + (size_t)q * m.out_ld;
  uint64_t *ol = m.out_label + (size_t)q * m.out_ld;
  for (uint32_t s2 = a.k + lane; s2 < m.out_ld; s2 += kWave) { od[s2] = __builtin_inff(); ol[s2] = kNoLabel; }
  uint32_t rank = 0;
  for (uint32_t t = 0; t < cnt; ++t) {
    const float td = readlane_f32(top.d[0], (int)t);
    const uint64_t tl = readlane_u64(top.lab[0], (int)t);
    rank += dl_less(td, tl, top.d[0], top.lab[0]) ? 1u : 0u;
  }
  if ((uint32_t)lane < cnt) { od[rank] = top.d[0]; ol[rank] = top.lab[0]; }
  if ((uint32_t)lane >= cnt && (uint32_t)lane < a.k) { od[lane] = __builtin_inff(); ol[lane] = kNoLabel; }
  if (lane == 0) m.out_n[q] = cnt;
  VK_STAMP(9);
#undef VK_STAMP
}

hipError_t launch_flat_rerank(const FlatScanArgs &a, const MergeArgs &m_in, bool l2, bool bf16, hipStream_t s) {
  if (a.nq == 0) return hipSuccess;
  if (a.k == 0 || a.k > 64 || a.cand_row == nullptr || a.done_cnt == nullptr || a.nrp != (uint32_t)kRerankParts || m_in.q_index || m_in.run_flag ||
      a.run_flag)
    return hipErrorInvalidValue;
  MergeArgs m = m_in;
  if (m.out_ld < a.k) m.out_ld = a.k;
  const size_t lds = std::max<size_t>((size_t)a.chunks * 64 + 4 * kRerankSlots * 4, ((size_t)3 * a.k + 2) * 12);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (a.cand_val != nullptr && (a.qcoef == nullptr || a.tile_norm == nullptr)) return hipErrorInvalidValue;
  const void *f = l2 ? (bf16 ? reinterpret_cast<const void *>(&flat_rerank_kernel<true, true>) : reinterpret_cast<const void *>(&flat_rerank_kernel<true, false>))
                     : (bf16 ? reinterpret_cast<const void *>(&flat_rerank_kernel<false, true>) : reinterpret_cast<const void *>(&flat_rerank_kernel<false, false>));
  if (lds > 48 * 1024) {
    hipError_t e = ensure_max_lds(f);
    if (e != hipSuccess) return e;
  }
  FlatScanArgs args = a;
  void *params[] = {&args, &m};
  return hipLaunchKernel(f, dim3(a.nq * kRerankParts), dim3(256), params, lds, s);
}

// Which query a merge block serves, and whether it has anything to do.  Plain launch: block b = query b.  Redo mode
// (MergeArgs::q_index): block b = compact query b < *nq_dev, lists at index b, output of query q_index[b].  A query the
// candidate filter handed over (ovf_q) is appended to the redo list and left to the exact pass.
__device__ __forceinline__ bool merge_block_query(const MergeArgs &a, uint64_t *q_in, uint64_t *q_out) {
  if (launch_skipped(a.run_flag, a.run_if, a.run_hi)) return false;
  uint64_t qi = blockIdx.x, qo = blockIdx.x;
  if (a.nq_dev) {
    if (qi >= *a.nq_dev) return false;
    qo = a.q_index[qi];
  }
  if (a.ovf_q && a.ovf_q[qo]) {
    if (threadIdx.x == 0) a.redo_list[atomicAdd(a.redo_cnt, 1u)] = (uint32_t)qo;
    return false;
  }
  *q_in = qi;
  *q_out = qo;
  return true;
}

// One wave per query: k best of `n_entries` (distance,label) pairs, written ascending.
// Empty entries are (+inf, kNoLabel).
template <int kE>
__global__ __launch_bounds__(256) void merge_topk_kernel(MergeArgs a) {
  // one block of four waves per query: each wave merges every fourth 256-entry slab into its own list, then
  // waves 1..3 hand their lists to wave 0 through LDS (a lone wave spent its time waiting for dependent loads)
  extern __shared__ float merge_lds[];
  uint64_t q, q_out;
  if (!merge_block_query(a, &q, &q_out)) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  WaveTopK<kE> top;
  top.init(a.k);
  for (uint32_t part = 0; part < a.parts; ++part) {
    const float *__restrict__ pd = a.in_dist + ((size_t)part * a.part_stride + q * a.q_stride);
    const uint64_t *__restrict__ pl = a.in_label + ((size_t)part * a.part_stride + q * a.q_stride);
    for (uint32_t i0 = (uint32_t)wave * 4 * kWave; i0 < a.per_part; i0 += 16 * kWave) {
      float dist[4];
      uint64_t lab[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kWave + lane;
        dist[u] = __builtin_inff();
        lab[u] = kNoLabel;
        if (i < a.per_part) { dist[u] = pd[i]; lab[u] = pl[i]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        uint64_t mask = __ballot(lab[u] != kNoLabel && dist[u] <= top.thr_d);
        while (mask) {
          const int b = __ffsll((unsigned long long)mask) - 1;
          mask &= mask - 1;
          const float cd = readlane_f32(dist[u], b);
          if (!(cd <= top.thr_d)) continue;
          top.insert(cd, readlane_u64(lab[u], b), lane);
        }
      }
    }
  }
  // waves 1..3 -> LDS -> wave 0
  const uint32_t cap = (uint32_t)kE * kWave;
  float *md = merge_lds;                                             // [3][cap]
  uint64_t *ml = reinterpret_cast<uint64_t *>(merge_lds + 3 * cap + (cap & 1));
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      md[(wave - 1) * cap + e * kWave + lane] = top.d[e];
      ml[(wave - 1) * cap + e * kWave + lane] = top.lab[e];
    }
  }
  __syncthreads();
  if (wave > 0) return;
  for (uint32_t i0 = 0; i0 < 3 * cap; i0 += kWave) {
    const float dist = md[i0 + lane];
    const uint64_t lab = ml[i0 + lane];
    uint64_t mask = __ballot(lab != kNoLabel && dist <= top.thr_d);
    while (mask) {
      const int b = __ffsll((unsigned long long)mask) - 1;
      mask &= mask - 1;
      const float cd = readlane_f32(dist, b);
      if (!(cd <= top.thr_d)) continue;
      top.insert(cd, readlane_u64(lab, b), lane);
    }
  }
  // rank sort of the kept entries (all keys distinct: labels are unique)
  const uint32_t cnt = top.cnt;
  float *od = a.out_dist + q_out * a.out_ld;
  uint64_t *ol = a.out_label + q_out * a.out_ld;
  for (uint32_t s = a.k + lane; s < a.out_ld; s += kWave) { od[s] = __builtin_inff(); ol[s] = kNoLabel; }
#pragma unroll
  for (int e = 0; e < kE; ++e) {
    const uint32_t s = (uint32_t)e * kWave + lane;
    uint32_t rank = 0;
#pragma unroll
    for (int e2 = 0; e2 < kE; ++e2) {
      for (int l2 = 0; l2 < kWave; ++l2) {
        const uint32_t t = (uint32_t)e2 * kWave + l2;
        if (t >= cnt) break;
        const float td = readlane_f32(top.d[e2], l2);
        const uint64_t tl = readlane_u64(top.lab[e2], l2);
        rank += dl_less(td, tl, top.d[e], top.lab[e]) ? 1u : 0u;
      }
    }
    if (s < cnt) { od[rank] = top.d[e]; ol[rank] = top.lab[e]; }
    if (s >= cnt && s < a.k) { od[s] = __builtin_inff(); ol[s] = kNoLabel; }
  }
  if (lane == 0) a.out_n[q_out] = cnt;
}

// Distances of an explicit row list (K8).  out[i] = distance(query, rows[idx[i]]).
template <bool kL2, bool kBf16>
__global__ __launch_bounds__(256) void gather_distance_kernel(GatherArgs a) {
  extern __shared__ float4 qs[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 3;
  const int rq = lane >> 2;
  const uint32_t chunks = a.chunks;
  for (uint32_t i = threadIdx.x; i < chunks * 4; i += blockDim.x)
    qs[i] = reinterpret_cast<const float4 *>(a.query)[i];
  __syncthreads();
  const uint32_t total_waves = gridDim.x * 4;
  const uint32_t n_tiles = (a.n + kRowsPerWave - 1) / kRowsPerWave;
  for (uint32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += total_waves) {
    const uint32_t i = tile * kRowsPerWave + rq;
    const bool valid = i < a.n;
    const uint32_t row = a.idx[valid ? i : a.n - 1];
    const float dist = quad_row_distance<kL2, kBf16>(row_base<kBf16>(a.rows, row, a.row_stride_f), qs, chunks, j);
    if (valid && j == 0) a.out[i] = dist;
  }
}

// The same merge by SELECTION, for k <= 64 and at most 4096 partial entries per query (every scan of a small or
// mid-sized index: the stream of inserts above is a serial chain of ~k*ln(n/k) wave-wide updates, 39 us for the 392
// lists of a 100k-row index, more than the scan itself).  Each thread keeps its <= 16 entries in registers; the block
// finds the k-th smallest distance key by a 32-step binary descent (a compare + ballot + s_bcnt1 per entry and step,
// one barrier per step), breaks a tie at that key by a second descent over the labels, compacts the exactly k winners
// into LDS and rank-sorts them.  Same answer as merge_topk_kernel: the k smallest by (distance, label), ascending.
__device__ __forceinline__ uint32_t merge_key(float f) {   // order-preserving f32 -> u32 (no NaNs reach a merge)
  uint32_t u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0;                            // -0 == +0 for the float compare of the other path
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int kSelPerThread = 16;
constexpr uint32_t kMergeSelectMax = 256 * kSelPerThread;

// Wave-wide selection over kN entries per lane (empty entries: key 0xFFFFFFFF, label kNoLabel): thresholds (T, L) such
// that exactly k entries satisfy key < T || (key == T && label <= L) -- the k smallest by (key, label); k <= the
// number of real entries.  Counts are ballots + s_bcnt1, wave-uniform: no LDS, no barrier.
template <int kN>
__device__ __forceinline__ void wave_select(const uint32_t (&key)[kN], const uint64_t (&lab)[kN], uint32_t k, uint32_t &T,
                                            uint64_t &L) {
  T = 0;
  for (int bit = 31; bit >= 0; --bit) {   // T = the k-th smallest key = the largest T with count(key < T) < k
    const uint32_t cand = T | (1u << bit);
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kN; ++u) c += (uint32_t)__popcll(__ballot(key[u] < cand));
    if (c < k) T = cand;
  }
  uint32_t below = 0, at = 0;
#pragma unroll
  for (int u = 0; u < kN; ++u) {
    below += (uint32_t)__popcll(__ballot(key[u] < T));
    at += (uint32_t)__popcll(__ballot(key[u] == T && lab[u] != kNoLabel));
  }
  L = ~0ull;
  if (below + at > k) {   // a tie at the k-th key: the k - below smallest labels among them
    const uint32_t need = k - below;
    L = 0;
    for (int bit = 63; bit >= 0; --bit) {
      const uint64_t cand = L | (1ull << bit);
      uint32_t c = 0;
#pragma unroll
      for (int u = 0; u < kN; ++u) c += (uint32_t)__popcll(__ballot(key[u] == T && lab[u] < cand));
      if (c < need) L = cand;
    }
  }
}

constexpr uint32_t kSelSurvivors = 256;   // 4 per lane of one wave

__global__ __launch_bounds__(256) void merge_select_kernel(MergeArgs a) {
  __shared__ uint32_t s_cnt[2][4];
  __shared__ uint32_t s_n, s_bound;
  __shared__ float c_d[64];
  __shared__ uint64_t c_l[64];
  __shared__ float v_d[kSelSurvivors];
  __shared__ uint64_t v_l[kSelSurvivors];
  uint64_t q, q_out;
  if (!merge_block_query(a, &q, &q_out)) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t total = a.parts * a.per_part;

  uint32_t key[kSelPerThread];
  uint64_t lab[kSelPerThread];
  float dv[kSelPerThread];
  {
    // all loads first (an entry past the end re-reads entry 0 and is dropped afterwards): one memory round trip
    const float *__restrict__ qd = a.in_dist + q * a.q_stride;
    const uint64_t *__restrict__ ql = a.in_label + q * a.q_stride;
    if (a.parts == 1) {
#pragma unroll
      for (int u = 0; u < kSelPerThread; ++u) {
        const uint32_t e = tid + 256u * u < total ? tid + 256u * u : 0u;
        lab[u] = ql[e];
        dv[u] = qd[e];
      }
    } else {   // segmented scans, shard merges: entry (part, i)
#pragma unroll
      for (int u = 0; u < kSelPerThread; ++u) {
        const uint32_t e = tid + 256u * u < total ? tid + 256u * u : 0u;
        const uint32_t part = e / a.per_part;
        const size_t o = (size_t)part * a.part_stride + (e - part * a.per_part);
        lab[u] = ql[o];
        dv[u] = qd[o];
      }
    }
#pragma unroll
    for (int u = 0; u < kSelPerThread; ++u) {
      if (tid + 256u * u >= total) lab[u] = kNoLabel;
      key[u] = lab[u] != kNoLabel ? merge_key(dv[u]) : 0xFFFFFFFFu;
    }
  }
  if (tid == 0) s_n = 0;

  uint32_t phase = 0;
  // block-wide sum of the waves' (uniform) counts through LDS, one barrier (the two halves of s_cnt alternate)
  auto block_sum = [&](uint32_t mine) -> uint32_t {
    if (lane == 0) s_cnt[phase][wave] = mine;
    __syncthreads();
    const uint32_t t = s_cnt[phase][0] + s_cnt[phase][1] + s_cnt[phase][2] + s_cnt[phase][3];
    phase ^= 1;
    return t;
  };
  uint32_t c = 0;
#pragma unroll
  for (int u = 0; u < kSelPerThread; ++u) c += (uint32_t)__popcll(__ballot(lab[u] != kNoLabel));
  const uint32_t real = block_sum(c);
  const uint32_t k = real < a.k ? real : a.k;
  float *od = a.out_dist + q_out * a.out_ld;
  uint64_t *ol = a.out_label + q_out * a.out_ld;
  for (uint32_t s = tid; s < a.out_ld; s += 256)
    if (s >= k) { od[s] = __builtin_inff(); ol[s] = kNoLabel; }
  if (tid == 0) a.out_n[q_out] = k;
  if (k == 0) return;

  // Fast path.  Wave 0 finds the k-th smallest key U of 256 of its entries (a sample; wave-local, no barriers);
  // the answer lies among the entries with key <= U.  If at most 256 survive (k up to about 15 for a full input),
  // wave 0 selects among them on its own: 4 barriers in all instead of one per bit.
  if (wave == 0) {
    uint32_t skey[4];
    uint64_t slab[4];
    uint32_t sreal = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      skey[u] = key[u];
      slab[u] = lab[u];
      sreal += (uint32_t)__popcll(__ballot(lab[u] != kNoLabel));
    }
    uint32_t U = 0xFFFFFFFFu;
    if (sreal >= k) {
      uint64_t Lu;
      wave_select<4>(skey, slab, k, U, Lu);
    }
    if (lane == 0) s_bound = U;
  }
  __syncthreads();
  const uint32_t U = s_bound;
  c = 0;
#pragma unroll
  for (int u = 0; u < kSelPerThread; ++u) c += (uint32_t)__popcll(__ballot(lab[u] != kNoLabel && key[u] <= U));
  const uint32_t survivors = block_sum(c);
  if (survivors <= kSelSurvivors) {
#pragma unroll
    for (int u = 0; u < kSelPerThread; ++u) {
      if (lab[u] != kNoLabel && key[u] <= U) {
        const uint32_t slot = atomicAdd(&s_n, 1u);
        v_d[slot] = dv[u];
        v_l[slot] = lab[u];
      }
    }
    __syncthreads();
    if (wave != 0) return;
    uint32_t key2[4];
    uint64_t lab2[4];
    float d2[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t i = lane + 64u * w;
      const bool have = i < survivors;
      d2[w] = have ? v_d[i] : __builtin_inff();
      lab2[w] = have ? v_l[i] : kNoLabel;
      key2[w] = have ? merge_key(d2[w]) : 0xFFFFFFFFu;
    }
    uint32_t T2;
    uint64_t L2;
    wave_select<4>(key2, lab2, k, T2, L2);
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const bool win = lab2[w] != kNoLabel && (key2[w] < T2 || (key2[w] == T2 && lab2[w] <= L2));
      const uint64_t m = __ballot(win);
      if (win) {
        const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        c_d[slot] = d2[w];
        c_l[slot] = lab2[w];
      }
      base += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes before its reads below
    if (lane < k) {
      const float md = c_d[lane];
      const uint64_t ml = c_l[lane];
      uint32_t rank = 0;
      for (uint32_t t = 0; t < k; ++t) rank += dl_less(c_d[t], c_l[t], md, ml) ? 1u : 0u;
      od[rank] = md;
      ol[rank] = ml;
    }
    return;
  }

  // General path: the same descent over all entries, block-wide (one barrier per bit).
  // T = the k-th smallest key: the largest T with count(key < T) < k
  uint32_t T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = T | (1u << bit);
    c = 0;
#pragma unroll
    for (int u = 0; u < kSelPerThread; ++u) c += (uint32_t)__popcll(__ballot(key[u] < cand));
    if (block_sum(c) < k) T = cand;
  }
  c = 0;
  uint32_t ce = 0;
#pragma unroll
  for (int u = 0; u < kSelPerThread; ++u) {
    c += (uint32_t)__popcll(__ballot(key[u] < T));
    ce += (uint32_t)__popcll(__ballot(key[u] == T && lab[u] != kNoLabel));
  }
  const uint32_t below = block_sum(c), at = block_sum(ce);
  // a tie at the k-th key: of the `at` entries there, the k - below with the smallest labels win
  uint64_t L = ~0ull;
  if (below + at > k) {
    const uint32_t need = k - below;
    L = 0;
    for (int bit = 63; bit >= 0; --bit) {
      const uint64_t cand = L | (1ull << bit);
      c = 0;
#pragma unroll
      for (int u = 0; u < kSelPerThread; ++u) c += (uint32_t)__popcll(__ballot(key[u] == T && lab[u] < cand));
      if (block_sum(c) < need) L = cand;
    }
  }
  // the k winners -> LDS (any order), then every winner's rank among them by (distance, label)
#pragma unroll
  for (int u = 0; u < kSelPerThread; ++u) {
    if (lab[u] != kNoLabel && (key[u] < T || (key[u] == T && lab[u] <= L))) {
      const uint32_t slot = atomicAdd(&s_n, 1u);
      c_d[slot] = dv[u];
      c_l[slot] = lab[u];
    }
  }
  __syncthreads();
  if (tid < k) {
    const float md = c_d[tid];
    const uint64_t ml = c_l[tid];
    uint32_t rank = 0;
    for (uint32_t t = 0; t < k; ++t) rank += dl_less(c_d[t], c_l[t], md, ml) ? 1u : 0u;
    od[rank] = md;
    ol[rank] = ml;
  }
}

// ---- launchers ----------------------------------------------------------------------------------
template <bool kL2, bool kBf16>
static hipError_t launch_scan_lb(const FlatScanArgs &a, dim3 grid, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    hipError_t e = ensure_max_lds(reinterpret_cast<const void *>(&flat_scan_kernel<1, kL2, 16, kBf16, true>));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((flat_scan_kernel<1, kL2, 16, kBf16, true>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

template <int kQB, bool kL2, int kE, bool kBf16>
static hipError_t launch_scan_t(const FlatScanArgs &a, dim3 grid, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    hipError_t e = ensure_max_lds(reinterpret_cast<const void *>(&flat_scan_kernel<kQB, kL2, kE, kBf16>));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((flat_scan_kernel<kQB, kL2, kE, kBf16>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

template <bool kL2, int kE, bool kBf16>
static hipError_t launch_scan_qb(int qb, const FlatScanArgs &a, dim3 grid, size_t lds, hipStream_t s) {
  switch (qb) {
    case 1: return launch_scan_t<1, kL2, kE, kBf16>(a, grid, lds, s);
    case 2: return launch_scan_t<2, kL2, kE, kBf16>(a, grid, lds, s);
    case 4: return launch_scan_t<4, kL2, kE, kBf16>(a, grid, lds, s);
    default: return launch_scan_t<8, kL2, kE, kBf16>(a, grid, lds, s);
  }
}

template <int kE, bool kBf16>
static hipError_t launch_scan_l2(bool l2, int qb, const FlatScanArgs &a, dim3 grid, size_t lds, hipStream_t s) {
  return l2 ? launch_scan_qb<true, kE, kBf16>(qb, a, grid, lds, s) : launch_scan_qb<false, kE, kBf16>(qb, a, grid, lds, s);
}

// re-rank launch: one query per block column (kQB = 1), k <= 1024 (kE = 1, 4 or 16), rows from the survivor lists
template <bool kL2, bool kBf16, int kE>
static hipError_t launch_rerank_t(const FlatScanArgs &a, dim3 grid, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    hipError_t e = ensure_max_lds(reinterpret_cast<const void *>(&flat_scan_kernel<1, kL2, kE, kBf16, false, true>));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((flat_scan_kernel<1, kL2, kE, kBf16, false, true>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}
template <int kE>
static hipError_t launch_rerank_e(const FlatScanArgs &a, bool l2, bool bf16, dim3 grid, size_t lds, hipStream_t s) {
  return l2 ? (bf16 ? launch_rerank_t<true, true, kE>(a, grid, lds, s) : launch_rerank_t<true, false, kE>(a, grid, lds, s))
            : (bf16 ? launch_rerank_t<false, true, kE>(a, grid, lds, s) : launch_rerank_t<false, false, kE>(a, grid, lds, s));
}

int flat_scan_slots_per_lane(uint64_t k) {
  if (k <= 64) return 1;
  if (k <= 256) return 4;
  if (k <= 1024) return 16;
  return 0;  // not served by the register-resident top-k
}

int flat_scan_pick_qb(uint64_t nq, uint32_t chunks, int e, bool l2) {
  // query block must fit LDS (<= 64 KiB so that two blocks share a CU) and, with wide
  // per-lane top-k state, registers
  static const int qb_cap = (int)VK_TUNE("VK_SCAN_QB", 8);
  int qb = nq >= 8 ? 8 : nq >= 4 ? 4 : nq >= 2 ? 2 : 1;
  if (qb > qb_cap) qb = qb_cap;
  // (occupancy decides here: with 8 row pieces in flight per lane, 8 queries per pass needed all 256 VGPRs for L2
  // -- one wave per SIMD, 798 QPS at 10Mx768 B=256; with 4 in flight it is 158 VGPRs, three waves, 2359 QPS)
  static const int qb_l2 = (int)VK_TUNE("VK_SCAN_QB_L2", 8);
  if (l2 && qb > qb_l2) qb = qb_l2;
  if (e > 1) qb = qb > 2 ? 2 : qb;
  if (e > 4) qb = 1;
  while (qb > 1 && (size_t)qb * chunks * 64 > 64 * 1024) qb >>= 1;
  return qb;
}

hipError_t launch_flat_scan(const FlatScanArgs &a, bool l2, bool bf16, int qb, int e, hipStream_t s) {
  if (a.nrp == 0 || (a.nrp & 7u) || a.nqg != (a.nq + qb - 1) / qb) return hipErrorInvalidValue;
  dim3 grid(a.nrp * a.nqg);
  size_t lds = (size_t)qb * a.chunks * 64;
  if (e == 1) lds = std::max<size_t>(lds, ((size_t)qb * 3 * a.k + 2) * 12);   // block-level merge buffers reuse it
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (a.cand_row) {  // re-rank of a survivor list
    if ((e != 1 && e != 4 && e != 16) || qb != 1) return hipErrorInvalidValue;
    return e == 1 ? launch_rerank_e<1>(a, l2, bf16, grid, lds, s)
                  : e == 4 ? launch_rerank_e<4>(a, l2, bf16, grid, lds, s) : launch_rerank_e<16>(a, l2, bf16, grid, lds, s);
  }
  if (a.lb_dist) {   // paged large-k scan
    if (e != 16 || qb != 1) return hipErrorInvalidValue;
    return l2 ? (bf16 ? launch_scan_lb<true, true>(a, grid, lds, s) : launch_scan_lb<true, false>(a, grid, lds, s))
              : (bf16 ? launch_scan_lb<false, true>(a, grid, lds, s) : launch_scan_lb<false, false>(a, grid, lds, s));
  }
  if (e == 1) return bf16 ? launch_scan_l2<1, true>(l2, qb, a, grid, lds, s) : launch_scan_l2<1, false>(l2, qb, a, grid, lds, s);
  if (e == 4) return bf16 ? launch_scan_l2<4, true>(l2, qb, a, grid, lds, s) : launch_scan_l2<4, false>(l2, qb, a, grid, lds, s);
  if (e == 16) return bf16 ? launch_scan_l2<16, true>(l2, qb, a, grid, lds, s) : launch_scan_l2<16, false>(l2, qb, a, grid, lds, s);
  return hipErrorInvalidValue;
}

hipError_t launch_merge_topk(const MergeArgs &a_in, int e, uint64_t nq, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  MergeArgs a = a_in;
  if (a.out_ld < a.k) a.out_ld = a.k;
  dim3 grid((uint32_t)nq);
  static const bool select = VK_TUNE("VK_MERGE_SELECT", 1) != 0;
  if (select && e == 1 && (uint64_t)a.parts * a.per_part <= kMergeSelectMax) {
    hipLaunchKernelGGL(merge_select_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const size_t lds = (size_t)(3 * e * 64 + 2) * 4 + (size_t)3 * e * 64 * 8;
  if (e == 1) hipLaunchKernelGGL((merge_topk_kernel<1>), grid, dim3(256), lds, s, a);
  else if (e == 4) hipLaunchKernelGGL((merge_topk_kernel<4>), grid, dim3(256), lds, s, a);
  else if (e == 16) hipLaunchKernelGGL((merge_topk_kernel<16>), grid, dim3(256), lds, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_gather_distance(const GatherArgs &a, bool l2, bool bf16, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  uint32_t tiles = (a.n + kRowsPerWave - 1) / kRowsPerWave;
  uint32_t blocks = (tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  size_t lds = (size_t)a.chunks * 64;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const void *f = l2 ? (bf16 ? reinterpret_cast<const void *>(&gather_distance_kernel<true, true>)
                             : reinterpret_cast<const void *>(&gather_distance_kernel<true, false>))
                     : (bf16 ? reinterpret_cast<const void *>(&gather_distance_kernel<false, true>)
                             : reinterpret_cast<const void *>(&gather_distance_kernel<false, false>));
  if (lds > 48 * 1024) {
    hipError_t e = ensure_max_lds(f);
    if (e != hipSuccess) return e;
  }
  GatherArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(f, dim3(blocks), dim3(256), params, lds, s);
}

// bound[q] as a float and as the order-preserving u32 key the shared per-query bound of K4 uses
__global__ void kth_bound_kernel(const float *out_dist, const uint32_t *out_n, uint32_t k, uint32_t nq, float *bound) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float b = out_n[q] >= k ? out_dist[(size_t)q * k + k - 1] : __builtin_inff();
  const uint32_t u = __float_as_uint(b);
  bound[q] = b;
  reinterpret_cast<uint32_t *>(bound)[nq + q] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void fill_empty_kernel(float *out_dist, uint64_t *out_label, uint32_t *out_n, uint32_t nq, uint32_t k) {
  const uint64_t total = (uint64_t)nq * k;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    out_dist[i] = __builtin_inff();
    out_label[i] = kNoLabel;
  }
  if (out_n == nullptr) return;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) out_n[q] = 0;
}
hipError_t launch_fill_empty(float *out_dist, uint64_t *out_label, uint32_t *out_n, uint32_t nq, uint32_t k, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  const uint64_t total = std::max<uint64_t>((uint64_t)nq * k, nq);
  hipLaunchKernelGGL(fill_empty_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 1024)), dim3(256), 0, s, out_dist,
                     out_label, out_n, nq, k);
  return hipGetLastError();
}

hipError_t launch_kth_bound(const float *out_dist, const uint32_t *out_n, uint32_t k, uint32_t nq, float *bound, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(kth_bound_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, out_dist, out_n, k, nq, bound);
  return hipGetLastError();
}

}  // namespace vk
