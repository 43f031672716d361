// hnsw_search.hip -- K5 + K6: HierarchicalNSW::searchKnn (third_party/hnswlib/hnswalg.h:1659-1725)
// on CDNA4: the greedy descent over the upper layers (:1667-1697) fused in front of the
// layer-0 best-first expansion searchBaseLayerST<false> (:351-551).  One 64-lane wave per
// query; throughput comes from thousands of queries in flight, each hop from 16 row
// gathers in flight per wave.
//
// What is kept from the reference, step for step:
//   * the entry point enters the result list only if it is live and allowed (:373-388)
//   * pop the closest frontier node; stop when it is farther than the worst result and the
//     result list holds ef entries (:400-413)
//   * its level-0 list is filtered against the visited set in list order (:453-464), every
//     unvisited neighbour gets a distance (:496), and the neighbours are then CONSIDERED in
//     list order with the bound updated after each one: consider = results.size() < ef ||
//     lowerBound > dist (:502-503); a considered node always joins the frontier, and joins
//     the results only if it is neither tombstoned nor filtered out (:506-524)
//   * results are trimmed to k and reported ascending by (distance,label) (:1715-1723,
//     vector_base.cc:258-277)
// What is organised differently (same answers, different machinery):
//   * distances: 16 neighbours at a time, one per quad of lanes, bit-identical to the
//     reference's SimSIMD order (device_common.hpp)
//   * visited set: one per resident wave in HBM instead of a u16 tag array -- a bitmap of the graph (atomicOr =
//     test-and-set) or, for full batches on large graphs, an exact open-addressing hash set of the ids a search
//     touches (atomicCAS); cleared by the wave at the start of each query
//   * result list: sorted ascending in registers across the lanes (ef <= 64*kE); worst = last
//   * frontier: unsorted pool in LDS with extract-min by a wave-wide scan; entries that can
//     never be popped (farther than the bound once the result list is full -- the bound
//     only shrinks from then on) are pruned when the pool fills
// Equal distances: hnswlib's heaps compare the distance only, so which of two equidistant
// nodes wins is decided by libstdc++'s sift order; here it is decided by arrival order.
// Ids can therefore differ from the CPU path only between exactly equidistant nodes.
#include <algorithm>
#include <type_traits>

#include <stdlib.h>

#include "device_common.hpp"
#include "kernels.hpp"


namespace vk {

namespace {

constexpr uint32_t kDeleteFlag = 0x00010000u;
constexpr uint32_t kNoneId = 0xFFFFFFFFu;
constexpr float kFltMax = 3.402823466e+38F;
constexpr int kVisBucketLog2 = 3, kVisBucket = 1 << kVisBucketLog2;   // ids per bucket of the visited table (vis_mode 2)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// vis_mode 3: the visited set in LDS -- LdsVis::kBuckets buckets of six 16-bit entries (12 B), 12 KB per wave.  An id
// below 2^24 has TWO home buckets, from two bijections of [0, 2^24) (odd multipliers): the top ten bits of the scrambled
// id name the bucket, the low fourteen are what is stored, next to one bit that says which bijection and one that says
// whether the entry sits in the home bucket or the one behind it (it goes there only when the home is full) -- so an
// entry names ONE id wherever it lies: the set is exact.  An id goes to the emptier of its homes (two choices keep the
// fullest bucket within a slot or two of the average); 0xFFFF = an empty slot; an id whose four candidate buckets are full
// lives in the wave's table in memory instead (hnsw_search_body).
// Two sizes: 1024 buckets of six entries (12 KB per wave, about 5500 ids before most new ones spill: two waves per SIMD), and
// -- one wave per SIMD with a whole row in flight per lane -- 2048 buckets of eight (32 KB; thirteen stored bits, eleven
// bucket bits).
template <bool kBig> struct LdsVis {
  static constexpr uint32_t kRemBits = kBig ? 13 : 14, kBuckets = kBig ? 2048 : 1024, kSlots = kBig ? 8 : 6;
  static constexpr uint32_t kWordsPerBucket = kSlots / 2, kWords = kBuckets * kWordsPerBucket;
};

// result list: rank r lives in slot r/64 of lane r%64, ascending by distance
template <int kE>
struct WaveSorted {
  float d[kE];
  uint32_t id[kE];
  uint32_t cnt;

  __device__ __forceinline__ void init() {
    cnt = 0;
#pragma unroll
    for (int e = 0; e < kE; ++e) { d[e] = __builtin_inff(); id[e] = kNoneId; }
  }
  __device__ __forceinline__ float at_d(uint32_t r) const {
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < kE; ++e)
      if ((r >> 6) == (uint32_t)e) v = readlane_f32(d[e], r & 63);
    return v;
  }
  __device__ __forceinline__ uint32_t at_id(uint32_t r) const {
    uint32_t v = 0;
#pragma unroll
    for (int e = 0; e < kE; ++e)
      if ((r >> 6) == (uint32_t)e) v = __builtin_amdgcn_readlane((int)id[e], r & 63);
    return v;
  }
  // insert (nd,nid) behind every entry with distance <= nd; the list keeps at most cap
  __device__ __forceinline__ void insert(float nd, uint32_t nid, uint32_t cap, int lane) {
    uint32_t pos = 0;
#pragma unroll
    for (int e = 0; e < kE; ++e)
      pos += __popcll(__ballot(((uint32_t)e * kWave + lane) < cnt && d[e] <= nd));
    if (pos >= cap) return;
#pragma unroll
    for (int e = kE - 1; e >= 0; --e) {
      if ((uint32_t)e * kWave + (kWave - 1) < pos) continue;  // whole slot row is in front of pos
      // lane i <- lane i-1 over the whole wave: DPP wave_shr:1, not a trip through the LDS crossbar (__shfl_up);
      // lane 0 is patched below
      float up_d = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[e]), 0x138, 0xF, 0xF, false));
      uint32_t up_id = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id[e], 0x138, 0xF, 0xF, false);
      float cd = 0.f;
      uint32_t cid = 0;
      if (e > 0) {
        cd = readlane_f32(d[e - 1], kWave - 1);
        cid = __builtin_amdgcn_readlane((int)id[e - 1], kWave - 1);
      }
      const uint32_t r = (uint32_t)e * kWave + lane;
      const float src_d = lane == 0 ? cd : up_d;
      const uint32_t src_id = lane == 0 ? cid : up_id;
      if (r > pos) { d[e] = src_d; id[e] = src_id; }
      else if (r == pos) { d[e] = nd; id[e] = nid; }
    }
    cnt = cnt + 1 < cap ? cnt + 1 : cap;
  }
};

// ef > 512: the result list does not fit the lanes' registers; it lives in LDS (one wave per block),
// sorted ascending like WaveSorted, insertion = parallel shift.  Same interface, lower throughput.
struct LdsSorted {
  float *d;
  uint32_t *id;
  uint32_t cnt;
  __device__ __forceinline__ void init() { cnt = 0; }
  __device__ __forceinline__ float at_d(uint32_t r) const { return d[r]; }
  __device__ __forceinline__ uint32_t at_id(uint32_t r) const { return id[r]; }
  __device__ __forceinline__ void insert(float nd, uint32_t nid, uint32_t cap, int lane) {
    uint32_t pos = 0;
    for (uint32_t base = 0; base < cnt; base += kWave) {
      const uint32_t i = base + lane;
      pos += __popcll(__ballot(i < cnt && d[i] <= nd));
    }
    if (pos >= cap) return;
    const uint32_t last = cnt < cap ? cnt : cap - 1;     // entries pos..last-1 move up by one
    for (uint32_t hi = last; hi > pos;) {
      const uint32_t i = hi - (uint32_t)lane;             // this lane's destination index
      const bool mv = hi >= (uint32_t)lane && i > pos;
      float v = 0.f;
      uint32_t w = 0;
      if (mv) { v = d[i - 1]; w = id[i - 1]; }
      __builtin_amdgcn_wave_barrier();
      if (mv) { d[i] = v; id[i] = w; }
      __builtin_amdgcn_wave_barrier();
      hi = hi > (uint32_t)kWave ? hi - kWave : 0;
      if (hi <= pos) break;
    }
    if (lane == 0) { d[pos] = nd; id[pos] = nid; }
    __builtin_amdgcn_wave_barrier();
    cnt = cnt + 1 < cap ? cnt + 1 : cap;
  }
};

// The frontier (hnswlib's candidate_set): an unsorted array with wave-wide extract-min.  kG = false: in LDS (the
// usual case: once the result list is full, entries beyond the bound are pruned and a few hundred slots suffice).
// kG = true: in HBM, for searches with a filter or tombstones (see HnswSearchArgs::pool_g); accesses bypass the
// CU's vector L1 (agent-scope relaxed atomics) because lane 0's appends must be seen by the other lanes' scans.
// (Plain 16-B loads behind an agent-scope release + acquire fence per hop were tried: 2.5x slower than this.)
template <bool kG>
struct Pool {
  float *d;
  uint32_t *id;
  uint32_t cnt, cap;
  __device__ __forceinline__ float ld_d(uint32_t i) const {
    if constexpr (kG) return __hip_atomic_load(d + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return d[i];
  }
  __device__ __forceinline__ uint32_t ld_id(uint32_t i) const {
    if constexpr (kG) return __hip_atomic_load(id + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return id[i];
  }
  __device__ __forceinline__ void st(uint32_t i, float dv, uint32_t iv) {
    if constexpr (kG) {
      __hip_atomic_store(d + i, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(id + i, iv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      d[i] = dv;
      id[i] = iv;
    }
  }
};

// drop entries that can never be expanded: farther than `bound`
template <bool kG>
__device__ __forceinline__ void pool_prune(Pool<kG> &c, float bound, int lane) {
  uint32_t w = 0;
  for (uint32_t base = 0; base < c.cnt; base += kWave) {
    const uint32_t i = base + lane;
    float dv = 0.f;
    uint32_t iv = 0;
    bool keep = false;
    if (i < c.cnt) { dv = c.ld_d(i); iv = c.ld_id(i); keep = !(dv > bound); }
    const uint64_t m = __ballot(keep);
    const uint32_t pos = w + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) c.st(pos, dv, iv);
    w += __popcll(m);
  }
  c.cnt = w;
}

}  // namespace

// Shared body.  kBatch = row pieces in flight per lane (device_common.hpp): 8 for the throughput kernel,
// 24 for the latency kernel that serves small batches.
// kGPool: where the frontier lives.  0 = LDS (no filter, no tombstones: it cannot outgrow 2*ef).  1 = HBM, at most 64k
// entries, one exact minimum per segment of 64 entries in LDS; a query whose frontier does not fit is ABANDONED and
// queued in a.redo_out.  2 = HBM, sized by the graph (a node enters the frontier at most once, so it cannot overflow),
// segment minima in HBM too and one minimum per 64 segments in LDS: the kernel that re-runs the abandoned queries.
template <bool kL2, int kE, bool kBf16, int kBatch, bool kSplitRows, int kGPool = 0, int kHash = 0>
__device__ __forceinline__ void hnsw_search_body(const HnswSearchArgs &a) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 3;
  const int rq = lane >> 2;
  const uint32_t chunks = a.chunks;

  // per-wave LDS carve: query | [result list d | id (kE == 0 only)] | pool d | pool id | nbr id | nbr dist
  constexpr bool kLdsList = kE == 0;
  const uint32_t list_words = kLdsList ? 2 * a.ef : 0;
  // (HBM frontier: the LDS keeps one minimum per segment of 64 entries in the pool's place, cand_cap / 64 floats;
  // two-level: one per 64 segments, cand_cap / 4096 floats)
  const uint32_t lds_pool = kGPool == 2 ? a.cand_cap / 8192 : kGPool == 1 ? a.cand_cap / 128 : a.cand_cap;
  // (kHash == 2: one count byte per bucket of the visited table, behind the neighbour arrays)
  const uint32_t vis_cnt_words = kHash == 5 ? LdsVis<true>::kWords : kHash == 3 ? LdsVis<false>::kWords : kHash == 2 ? (1u << a.vis_hash_log2) / (kVisBucket * 4u) : 0u;
  const size_t per_wave_f4 = (size_t)chunks * 4 + (list_words + lds_pool * 2 + a.nbr_cap * 2 + vis_cnt_words + 3) / 4;
  float4 *qs = lds4 + wave * per_wave_f4;
  float *list_d = reinterpret_cast<float *>(qs + chunks * 4);
  float *pool_d = list_d + list_words;
  uint32_t *pool_id = reinterpret_cast<uint32_t *>(pool_d + lds_pool);
  uint32_t *nbr_id = pool_id + lds_pool;
  float *nbr_d = reinterpret_cast<float *>(nbr_id + a.nbr_cap);
  uint32_t *vis_cnt = reinterpret_cast<uint32_t *>(nbr_d + a.nbr_cap);

  const uint32_t wpb = blockDim.x >> 6;                  // 4 waves per block, 1 with the LDS result list
  const uint32_t wslot = blockIdx.x * wpb + wave;
  const uint32_t wstride = gridDim.x * wpb;
  uint32_t *bitmap = a.visited + (size_t)wslot * a.bitmap_words;

  unsigned long long st_eval = 0, st_hops = 0, st_over = 0, st_q = 0;

  // every wave slot starts on query `wslot`; further queries are handed out first come, first served (a.queue), so
  // a batch that is not a multiple of the resident waves, or whose queries differ in length, still ends together.
  // With a.redo_in the work list is that array (the queries an earlier launch abandoned): [0] = count, then ids.
  const uint32_t n_work = a.redo_in ? __hip_atomic_load(a.redo_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.nq;
  uint32_t qi = wslot;
  while (qi < n_work) {
    const uint32_t q = a.redo_in ? a.redo_in[1 + qi] : qi;
    unsigned long long q_eval = 0, q_hops = 0;
    uint32_t q_vis = 0;            // kHash: entries in the visited table
    bool abandoned = false;
    // this query's filter: its own bitmap when the batch carries one per query (InlineVectorFilter is built per
    // FT.SEARCH, search.cc:103-134), else the batch's
    const uint64_t *q_bits = a.allow_tab ? a.allow_tab[q] : a.allow_bits;
    const uint64_t q_nbits = a.allow_tab ? a.allow_nbits_tab[q] : a.allow_nbits;
    if (poll_cancel(a.cancel) || (a.cancel_q && poll_cancel(a.cancel_q + q))) {   // cancelled before this query started: an empty answer, and on to drain the queue
      for (uint32_t r = lane; r < a.k; r += kWave) { a.out_dist[(size_t)q * a.k + r] = __builtin_inff(); a.out_label[(size_t)q * a.k + r] = kNoLabel; }
      if (lane == 0) a.out_n[q] = 0;
      uint32_t nxt0 = 0;
      if (lane == 0) nxt0 = atomicAdd(a.queue, 1u);
      qi = wstride + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt0);
      continue;
    }
    // ---- stage the query, clear this wave's visited bitmap --------------------------------
    {
      const float4 *src = reinterpret_cast<const float4 *>(a.queries + (size_t)q * a.q_stride_f);
      for (uint32_t i = lane; i < chunks * 4; i += kWave) qs[i] = src[i];
      if constexpr (kHash == 3 || kHash == 5) {   // every slot empty
        for (uint32_t i = lane; i < vis_cnt_words; i += kWave) vis_cnt[i] = 0xFFFFFFFFu;
      } else if constexpr (kHash == 2) {   // the counts say which table words mean anything: the table itself is never cleared
        for (uint32_t i = lane; i < vis_cnt_words; i += kWave) vis_cnt[i] = 0;
      } else {
        uint4 *bm4 = reinterpret_cast<uint4 *>(bitmap);
        const uint4 z = kHash ? make_uint4(kNoneId, kNoneId, kNoneId, kNoneId) : make_uint4(0, 0, 0, 0);
        for (uint32_t i = lane; i < a.bitmap_words / 4; i += kWave) bm4[i] = z;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    const bool can_split = kSplitRows && chunks % kBatch == 0 && chunks >= 2 * kBatch;
    auto row_dist = [&](uint32_t id) -> float {
      return quad_row_distance<kL2, kBf16, kBatch>(row_base<kBf16>(a.rows, id, a.row_stride_f), qs, chunks, j);
    };
    // kHash == 2 -- the visited set without atomics on memory.  The table is `buckets` buckets of kVisBucket ids (32 B)
    // in HBM; how many ids a bucket holds is a byte in LDS.  An id lives in the first bucket from its home bucket on that
    // had room when it came (buckets only fill up, so every bucket in front of it is full for good).  Look-up: the
    // count from LDS -- zero: the id is new and nothing is read -- else ONE 32-B read of the bucket, on to the next
    // bucket only behind a full one.  Insert: a slot number from an LDS atomic (the lanes of the wave insert the distinct
    // ids of one list at the same time) and a 4-B store nobody waits for.  Exact like the table of mode 0/1: ids, not
    // fingerprints.  A count can pass kVisBucket only by the lanes that raced for the last slots (< 64): it fits its byte.
    const uint32_t vis_bmask = ((1u << a.vis_hash_log2) / kVisBucket) - 1u;
    const __amdgpu_buffer_rsrc_t vis_rsrc = __builtin_amdgcn_make_buffer_rsrc(bitmap, 0, (int)(a.bitmap_words * 4u), 0x00020000);
    auto vis_count = [&](uint32_t b) -> uint32_t { return (vis_cnt[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu; };
    auto vis_insert = [&](uint32_t id, uint32_t b) {   // b: a bucket the id's probe sequence has reached
      for (;;) {
        if (vis_count(b) < (uint32_t)kVisBucket) {
          const uint32_t sh = (b & 3u) * 8u;
          const uint32_t slot = (atomicAdd(&vis_cnt[b >> 2], 1u << sh) >> sh) & 0xFFu;
          if (slot < (uint32_t)kVisBucket) {
            __builtin_amdgcn_raw_buffer_store_b32(id, vis_rsrc, (int)((b * kVisBucket + slot) * 4u), 0, 16);
            return;
          }
        }
        b = (b + 1) & vis_bmask;
      }
    };
    auto vis_lookup = [&](uint32_t id, uint32_t &b) -> bool {   // true: visited before; false: new, b = where to insert from
      b = (id * 2654435761u) >> (32u - a.vis_hash_log2 + kVisBucketLog2);
      for (;;) {
        const uint32_t c = vis_count(b);
        if (c) {
          // (sc1: past this CU's vector L1 -- the stores above are not looked for there)
          const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(vis_rsrc, (int)(b * kVisBucket * 4u), 0, 16);
          bool hit = (c > 0 && v0[0] == id) || (c > 1 && v0[1] == id) || (c > 2 && v0[2] == id) || (c > 3 && v0[3] == id);
          if (c > 4) {
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(vis_rsrc, (int)(b * kVisBucket * 4u + 16u), 0, 16);
            hit = hit || v1[0] == id || (c > 5 && v1[1] == id) || (c > 6 && v1[2] == id) || (c > 7 && v1[3] == id);
          }
          if (hit) return true;
        }
        if (c < (uint32_t)kVisBucket) return false;
        b = (b + 1) & vis_bmask;
      }
    };
    // kHash == 3 / 5 -- the set in LDS (see LdsVis).  Look-up and insert are one walk over at most four buckets: the id's
    // entry found -> visited; a free slot -> taken with a compare-and-swap of its 32-bit word (the lanes of the wave insert
    // the ids of one list at the same time; one that loses the word reads the bucket again -- and finds its own id there if
    // the list named it twice).  An id whose four candidate buckets are all full goes to the wave's table in HBM instead
    // (the compare-and-swap table of mode 0, cleared when a query first needs it): full buckets stay full, so the same id
    // finds them full again next time and is looked up THERE -- no id is ever in both places, nothing is re-run, and a
    // search that outgrows the on-chip set merely pays memory accesses for its last few hundred ids.
    auto visit_lds = [&](uint32_t id) -> uint32_t {   // 0 = visited before, 1 = new (now in the set), 2 = new to the set, but its four buckets are full
      using LV = LdsVis<kHash == 5>;
      constexpr uint32_t kW = LV::kWordsPerBucket;
      // two bijections of [0, 2^24): two home buckets (top bits) with their remainders (low bits)
      const uint32_t H0 = (id * 0x9E3779B1u) & 0xFFFFFFu, H1 = (id * 0x85EBCA6Bu + 0x5BD1E9u) & 0xFFFFFFu;
      for (;;) {
        uint32_t best_free = 0, best_at = 0, best_old = 0, best_new = 0;
#pragma unroll
        for (uint32_t c = 0; c < 2; ++c) {
          const uint32_t Hc = c ? H1 : H0;
          for (uint32_t d = 0; d < 2; ++d) {
            const uint32_t at = (((Hc >> LV::kRemBits) + d) & (LV::kBuckets - 1u)) * kW;
            const uint32_t key = (c << (LV::kRemBits + 1)) | (d << LV::kRemBits) | (Hc & ((1u << LV::kRemBits) - 1u));
            // (the one 16-bit key that reads as "empty" is never stored: an id cannot live in the bucket it would need it
            //  for -- not looked for there (an empty slot would answer "visited"), not put there)
            if (key == 0xFFFFu) continue;
            uint32_t w[kW];
#pragma unroll
            for (uint32_t t = 0; t < kW; ++t) w[t] = vis_cnt[at + t];
            bool hit = false;
            uint32_t nfree = 0;
#pragma unroll
            for (uint32_t t = 0; t < kW; ++t) {
              hit = hit || (w[t] & 0xFFFFu) == key || (w[t] >> 16) == key;
              nfree += ((w[t] & 0xFFFFu) == 0xFFFFu) + ((w[t] >> 16) == 0xFFFFu);
            }
            if (hit) return 0u;
            if (nfree == 0) continue;                         // full: the id may sit one bucket further
            if (nfree > best_free) {
              // entries fill a bucket front to back: the first free slot is number `used`
              const uint32_t used = LV::kSlots - nfree, wi = used >> 1;
              uint32_t old = w[0];
#pragma unroll
              for (uint32_t t = 1; t < kW; ++t) old = wi == t ? w[t] : old;
              best_free = nfree;
              best_at = at + wi;
              best_old = old;
              best_new = (used & 1u) ? (old & 0xFFFFu) | (key << 16) : (old & 0xFFFF0000u) | key;
            }
            break;                                             // a bucket with room never sent anything further
          }
        }
        if (best_free == 0) return 2u;
        if (atomicCAS(&vis_cnt[best_at], best_old, best_new) == best_old) return 1u;
      }
    };
    auto visit_hbm = [&](uint32_t id) -> bool {   // the table in memory: exact set of ids, linear probing, compare-and-swap
      const uint32_t mask = (1u << a.vis_hash_log2) - 1u;
      uint32_t h = (id * 2654435761u) >> (32u - a.vis_hash_log2);
      for (;;) {
        const uint32_t old = atomicCAS(&bitmap[h], kNoneId, id);
        if (old == kNoneId) return true;
        if (old == id) return false;
        h = (h + 1) & mask;
      }
    };
    bool hbm_used = false;       // kHash 3 / 5: this query has spilled ids into the table in memory (cleared on first use)
    uint32_t q_hbm = 0;          // ... how many
    auto visit = [&](uint32_t id) -> bool {  // true if it was NOT visited before
      if constexpr (kHash == 3 || kHash == 5) {
        return visit_lds(id) == 1u;                          // (the entry point: an empty set has room)
      } else if constexpr (kHash == 2) {
        uint32_t b;
        if (vis_lookup(id, b)) return false;
        vis_insert(id, b);
        return true;
      } else if constexpr (kHash == 1) {
        // exact set of ids, linear probing from a multiplicative hash; never more than 3/4 full (checked per hop), so
        // the probe ends.  The lanes of the wave insert the (distinct) ids of one list at the same time: two that meet in
        // a slot are told apart by the compare-and-swap
        const uint32_t mask = (1u << a.vis_hash_log2) - 1u;
        uint32_t h = (id * 2654435761u) >> (32u - a.vis_hash_log2);
        for (;;) {
          uint32_t old;
          if (a.vis_mode == 1) {
            old = kNoneId;
            (void)__hip_atomic_compare_exchange_strong(&bitmap[h], &old, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          } else {
            old = atomicCAS(&bitmap[h], kNoneId, id);
          }
          if (old == kNoneId) return true;
          if (old == id) return false;
          h = (h + 1) & mask;
        }
      } else {
        const uint32_t bit = 1u << (id & 31);
        return (atomicOr(&bitmap[id >> 5], bit) & bit) == 0;
      }
    };

    // ---- K6: greedy descent over the upper layers (:1667-1697) -------------------------------
    uint32_t cur = a.entry_point;
    float curdist = readlane_f32(row_dist(cur), 0);
    for (int level = a.max_level; level > 0; --level) {
      bool changed = true;
      while (changed) {
        changed = false;
        const uint32_t *ll = a.upper_pool + (size_t)(a.upper_slot[cur] + (uint32_t)(level - 1)) * a.up_stride;
        const uint32_t size = ll[0] & 0xFFFFu;
        for (uint32_t base = 0; base < size; base += kRowsPerWave) {
          const uint32_t i = base + rq;
          const bool valid = i < size;
          const uint32_t cid = valid ? ll[1 + i] : cur;
          float dv = row_dist(cid);
          // sequential "d < curdist" over the list == first occurrence of the minimum
          float bd = valid ? dv : __builtin_inff();
          uint32_t bi = i;
#pragma unroll
          for (int m = 4; m < kWave; m <<= 1) {
            const float od = __shfl_xor(bd, m);
            const uint32_t oi = __shfl_xor((int)bi, m);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
          }
          if (bd < curdist) {
            curdist = bd;
            cur = ll[1 + bi];
            changed = true;
          }
        }
      }
    }

    // ---- K5: layer-0 best-first expansion (:351-551) --------------------------------------------
    typename std::conditional<kLdsList, LdsSorted, WaveSorted<(kE > 0 ? kE : 1)>>::type top;
    if constexpr (kLdsList) { top.d = list_d; top.id = reinterpret_cast<uint32_t *>(list_d + a.ef); }
    top.init();
    Pool<(kGPool != 0)> c{pool_d, pool_id, 0, a.cand_cap};
    // kGPool: seg_min[s] = min distance of entries 64s .. 64s+63, exact at all times (LDS).  kGPool == 2: seg_min lives
    // in HBM behind the entries (agent-scope accesses like them) and sup_min[t] = min of seg_min[64t .. 64t+63] in LDS
    float *seg_min = pool_d;
    float *sup_min = pool_d;
    float tail_min = 0.f;      // kGPool == 2: minimum of the segment the next append goes to (wave-uniform)
    if constexpr (kGPool != 0) {
      const size_t per_wave = kGPool == 2 ? (size_t)2 * a.cand_cap + a.cand_cap / kWave : (size_t)2 * a.cand_cap;
      c.d = a.pool_g + (size_t)wslot * per_wave;
      c.id = reinterpret_cast<uint32_t *>(c.d + a.cand_cap);
      if constexpr (kGPool == 2) seg_min = c.d + 2 * (size_t)a.cand_cap;
    }
    auto ld_seg = [&](uint32_t sidx) -> float {
      if constexpr (kGPool == 2) return __hip_atomic_load(seg_min + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else return seg_min[sidx];
    };
    auto st_seg = [&](uint32_t sidx, float v) {
      if constexpr (kGPool == 2) __hip_atomic_store(seg_min + sidx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else seg_min[sidx] = v;
    };
    // all minima from the entries (after a compaction)
    auto rebuild_minima = [&]() {
      if constexpr (kGPool != 0) {
        const uint32_t nseg = (c.cnt + kWave - 1) / kWave;
        float gmin = __builtin_inff();
        for (uint32_t sidx = 0; sidx < nseg; ++sidx) {
          const uint32_t i = sidx * kWave + lane;
          const float mv = wave_min_f32(i < c.cnt ? c.ld_d(i) : __builtin_inff());
          if (lane == 0) st_seg(sidx, mv);
          tail_min = mv;
          if constexpr (kGPool == 2) {
            gmin = (sidx % kWave) == 0 ? mv : fminf(gmin, mv);
            if (lane == 0 && ((sidx % kWave) == kWave - 1 || sidx + 1 == nseg)) sup_min[sidx / kWave] = gmin;
          }
        }
      }
    };
    float lowerBound;
    {
      bool ep_ok = true;
      if (a.check_deleted && (a.links0[(size_t)cur * a.l0_stride] & kDeleteFlag)) ep_ok = false;
      if (ep_ok && q_bits && !allow_bit(q_bits, q_nbits, a.labels[cur])) ep_ok = false;
      const float d0 = ep_ok ? curdist : kFltMax;
      if (ep_ok) {
        lowerBound = curdist;   // the reference recomputes the same distance (:378)
        top.insert(curdist, cur, a.ef, lane);
        q_eval += 1;
      } else {
        lowerBound = kFltMax;
      }
      if (lane == 0) {
        c.st(0, d0, cur);
        if constexpr (kGPool != 0) st_seg(0, d0);
        if constexpr (kGPool == 2) sup_min[0] = d0;
      }
      tail_min = d0;
      c.cnt = 1;
      if (lane == 0) (void)visit(cur);
      q_vis = 1;
    }

    for (;;) {
      if (c.cnt == 0 || abandoned) break;
      if ((a.cancel || a.cancel_q) && (q_hops % kCancelPollHops) == kCancelPollHops - 1 &&
          (poll_cancel(a.cancel) || (a.cancel_q && poll_cancel(a.cancel_q + q)))) break;   // :400-402
      // extract-min over the pool
      float bd = __builtin_inff();
      uint32_t bi = kNoneId;
      float seg_v = 0.f;        // kGPool: this lane's entry of the winning segment
      float grp_v = 0.f;        // kGPool == 2: this lane's segment minimum in the winning group of 64 segments
      uint32_t seg_s = 0;
      if constexpr (kGPool != 0) {
        // the minimum over the per-segment minima, then one load per lane of the winning segment: the entry is the
        // first one there that equals it -- the smallest pool index among equal distances, like the scan below
        uint32_t bs = kNoneId;
        const uint32_t nseg = (c.cnt + kWave - 1) / kWave;
        const uint32_t ntop = kGPool == 2 ? (nseg + kWave - 1) / kWave : nseg;
        const float *top_min = kGPool == 2 ? sup_min : seg_min;
        for (uint32_t sidx = lane; sidx < ntop; sidx += kWave) {
          const float v = top_min[sidx];
          if (v < bd || bs == kNoneId) { bd = v; bs = sidx; }
        }
        {   // (the distance alone first, like the LDS frontier below)
          const float md = wave_min_f32(bd);
          const uint64_t at_min = __ballot(bs != kNoneId && bd == md);
          if (__popcll(at_min) == 1) {
            bs = (uint32_t)__builtin_amdgcn_readlane((int)bs, __ffsll((unsigned long long)at_min) - 1);
            bd = md;
          } else {
#pragma unroll
            for (int m = 1; m < kWave; m <<= 1) {
              const float od = __shfl_xor(bd, m);
              const uint32_t os = __shfl_xor((int)bs, m);
              if (os != kNoneId && (bs == kNoneId || od < bd || (od == bd && os < bs))) { bd = od; bs = os; }
            }
          }
        }
        if constexpr (kGPool == 2) {   // one level down: the first segment of that group whose minimum it is
          const uint32_t si = bs * kWave + lane;
          grp_v = si < nseg ? ld_seg(si) : __builtin_inff();
          const uint64_t hit1 = __ballot(si < nseg && grp_v == bd);
          bs = bs * kWave + (uint32_t)(__ffsll((unsigned long long)hit1) - 1);
        }
        seg_s = bs;
        const uint32_t i = seg_s * kWave + lane;
        seg_v = i < c.cnt ? c.ld_d(i) : __builtin_inff();
        const uint64_t hit = __ballot(i < c.cnt && seg_v == bd);
        bi = seg_s * kWave + (uint32_t)(__ffsll((unsigned long long)hit) - 1);
      } else {
        for (uint32_t i = lane; i < c.cnt; i += kWave) {
          const float dv = c.ld_d(i);
          if (dv < bd || bi == kNoneId) { bd = dv; bi = i; }
        }
        // the minimum distance alone first (a DPP reduction, not a ds_bpermute per step and value); only when several
        // lanes hold it do the indices take part -- the smallest pool index among equal distances wins
        const float md = wave_min_f32(bd);
        const uint64_t at_min = __ballot(bi != kNoneId && bd == md);
        if (__popcll(at_min) == 1) {
          bi = (uint32_t)__builtin_amdgcn_readlane((int)bi, __ffsll((unsigned long long)at_min) - 1);
          bd = md;
        } else {
#pragma unroll
          for (int m = 1; m < kWave; m <<= 1) {
            const float od = __shfl_xor(bd, m);
            const uint32_t oi = __shfl_xor((int)bi, m);
            if (oi != kNoneId && (bi == kNoneId || od < bd || (od == bd && oi < bi))) { bd = od; bi = oi; }
          }
        }
      }
      const float cand_dist = bd;
      if (cand_dist > lowerBound && top.cnt == a.ef) break;
      const uint32_t cur_id = c.ld_id(bi);
      // remove: move the last entry into the hole
      if constexpr (kGPool != 0) {
        const uint32_t last = c.cnt - 1, sl = last / kWave;
        const float d_last = c.ld_d(last);
        const uint32_t id_last = c.ld_id(last);
        if (lane == 0 && bi != last) c.st(bi, d_last, id_last);
        // the two segments touched: the hole's (its values are in seg_v) and the last entry's
        float nv = seg_v;
        if (seg_s * kWave + lane == bi) nv = bi != last ? d_last : __builtin_inff();
        if (seg_s == sl && seg_s * kWave + lane == last) nv = __builtin_inff();
        const float m0 = wave_min_f32(nv);
        float m1 = m0;
        if (seg_s != sl) {
          const uint32_t i = sl * kWave + lane;
          m1 = wave_min_f32(i < last ? c.ld_d(i) : __builtin_inff());
          if (lane == 0) st_seg(sl, m1);
        }
        if (lane == 0) st_seg(seg_s, m0);
        tail_min = m1;          // the segment the next append goes to is the last entry's (or a fresh one)
        if constexpr (kGPool == 2) {
          // the group minima above them, from the values this lane already holds (no re-read of what was just stored)
          const uint32_t g0 = seg_s / kWave, g1 = sl / kWave;
          float gv = grp_v;
          if ((uint32_t)lane == seg_s % kWave) gv = m0;
          if (g1 == g0 && (uint32_t)lane == sl % kWave) gv = m1;
          const float n0 = wave_min_f32(gv);
          if (lane == 0) sup_min[g0] = n0;
          if (g1 != g0) {
            const uint32_t nseg = (c.cnt + kWave - 1) / kWave;
            const uint32_t si = g1 * kWave + lane;
            float v = si < nseg ? ld_seg(si) : __builtin_inff();
            if ((uint32_t)lane == sl % kWave) v = m1;
            const float n1 = wave_min_f32(v);
            if (lane == 0) sup_min[g1] = n1;
          }
        }
      } else {
        if (lane == 0 && bi != c.cnt - 1) c.st(bi, c.ld_d(c.cnt - 1), c.ld_id(c.cnt - 1));
      }
      c.cnt -= 1;
      q_hops += 1;

      // phase 1: unvisited neighbours, list order preserved
      const uint32_t *ll = a.links0 + (size_t)cur_id * a.l0_stride;
      // (the first 64 ids are requested together with the count in front of them, not behind it: one trip to memory;
      // slots past the count hold whatever the list held before -- never used)
      const uint32_t nid_first = (uint32_t)lane + 1 < a.l0_stride ? ll[1 + lane] : 0u;
      const uint32_t size = ll[0] & 0xFFFFu;
      if constexpr (kHash == 3 || kHash == 5) {   // (only the spill table can fill up)
        if (q_hbm + size > (3u << a.vis_hash_log2) / 4u) { abandoned = true; break; }
      } else if constexpr (kHash != 0) {   // the table must not fill up: this query goes to the launch with the bitmap
        if (q_vis + size > (3u << a.vis_hash_log2) / 4u) { abandoned = true; break; }
      }
      uint32_t nn = 0;
      for (uint32_t base = 0; base < size; base += kWave) {
        const uint32_t i = base + lane;
        bool unv = false;
        uint32_t nid = 0;
        if constexpr (kHash == 3 || kHash == 5) {
          uint32_t r = 0;
          if (i < size) { nid = base == 0 ? nid_first : ll[1 + i]; r = visit_lds(nid); }
          unv = r == 1u;
          const uint64_t spill = __ballot(r == 2u);
          if (spill != 0) {                                   // rare: ids without room on chip
            if (!hbm_used) {
              uint4 *bm4 = reinterpret_cast<uint4 *>(bitmap);
              for (uint32_t t = lane; t < a.bitmap_words / 4; t += kWave) bm4[t] = make_uint4(kNoneId, kNoneId, kNoneId, kNoneId);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              hbm_used = true;
            }
            if (r == 2u) unv = visit_hbm(nid);
            q_hbm += (uint32_t)__popcll(spill);
          }
        } else {
          if (i < size) { nid = base == 0 ? nid_first : ll[1 + i]; unv = visit(nid); }
        }
        const uint64_t m = __ballot(unv);
        if (unv) nbr_id[nn + __popcll(m & ((1ull << lane) - 1ull))] = nid;
        nn += __popcll(m);
      }
      // phase 3a: distances, 16 rows per round; a round of <= 8 (<= 4) rows gives every row two (four) quads, which
      // fetch alternate batches of its pieces (same arithmetic, half / a quarter of the memory round trips)
      for (uint32_t base = 0; base < nn;) {
        const uint32_t rem = nn - base;
        if (rem > 8 || !can_split) {
          const uint32_t u = base + rq;
          const uint32_t nid = nbr_id[u < nn ? u : nn - 1];
          const float dv = row_dist(nid);
          if (u < nn && j == 0) nbr_d[u] = dv;
          base += kRowsPerWave;
        } else if (rem > 4) {
          const uint32_t u = base + ((lane & 31) >> 2);
          const uint32_t nid = nbr_id[u < nn ? u : nn - 1];
          const float dv = quad_row_distance_split<kL2, kBf16, kBatch, 2>(row_base<kBf16>(a.rows, nid, a.row_stride_f), qs, chunks, lane);
          if (u < nn && (lane & 35) == 0) nbr_d[u] = dv;      // j == 0 of group 0
          base += 8;
        } else {
          const uint32_t u = base + ((lane & 15) >> 2);
          const uint32_t nid = nbr_id[u < nn ? u : nn - 1];
          const float dv = quad_row_distance_split<kL2, kBf16, kBatch, 4>(row_base<kBf16>(a.rows, nid, a.row_stride_f), qs, chunks, lane);
          if (u < nn && (lane & 51) == 0) nbr_d[u] = dv;
          base += 4;
        }
      }
      q_eval += nn;
      q_vis += nn;
      // phase 3b: consider in list order, bound updated after each neighbour
      for (uint32_t base = 0; base < nn && !abandoned; base += kWave) {
        const uint32_t u = base + lane;
        const float dv = u < nn ? nbr_d[u] : __builtin_inff();
        const uint32_t nid = u < nn ? nbr_id[u] : 0;
        // once the list is full the bound only shrinks: anything not below it now never will be
        uint64_t m = __ballot(u < nn && (top.cnt < a.ef || lowerBound > dv));
        // tombstone and filter of every neighbour that may be considered, looked up by its own lane BEFORE the walk in
        // list order: inside it they were two dependent loads (label, bitmap word) per neighbour, one neighbour at a time
        // (hybrid shard, same lease: 31.2k -> 31.7k QPS)
        uint64_t okm = ~0ull;
        if (a.check_deleted || q_bits) {
          bool okl = true;
          if ((m >> lane) & 1ull) {
            if (a.check_deleted && (a.links0[(size_t)nid * a.l0_stride] & kDeleteFlag)) okl = false;
            if (okl && q_bits && !allow_bit(q_bits, q_nbits, a.labels[nid])) okl = false;
          }
          okm = __ballot(okl);
        }
        while (m) {
          const int b = __ffsll((unsigned long long)m) - 1;
          m &= m - 1;
          const float cd = readlane_f32(dv, b);
          if (!(top.cnt < a.ef || lowerBound > cd)) continue;
          const uint32_t cid = __builtin_amdgcn_readlane((int)nid, b);
          // frontier
          if (c.cnt == c.cap) {
            if (top.cnt == a.ef) {
              pool_prune(c, lowerBound, lane);
              rebuild_minima();   // the compaction moved everything
            }
            if (c.cnt == c.cap) {
              // The reference's candidate_set is unbounded (hnswalg.h:367-370,506): dropping the entry would end the
              // search early.  The capped HBM frontier gives the query up instead -- it is re-run by the launch with the
              // graph-sized frontier (a.redo_out); the LDS frontier and the graph-sized one cannot get here
              // (counted, and the host turns a non-zero count into an error).
              if ((kGPool == 1 || kHash) && a.redo_out) { abandoned = true; break; }
              st_over += 1;
            }
          }
          if (c.cnt < c.cap) {
            if constexpr (kGPool == 2) {
              tail_min = (c.cnt % kWave) == 0 ? cd : fminf(tail_min, cd);
              if (lane == 0) {
                c.st(c.cnt, cd, cid);
                st_seg(c.cnt / kWave, tail_min);
                const uint32_t g = c.cnt / (kWave * kWave);
                sup_min[g] = (c.cnt % (kWave * kWave)) == 0 ? cd : fminf(sup_min[g], cd);
              }
            } else if (lane == 0) {
              c.st(c.cnt, cd, cid);
              if constexpr (kGPool == 1) {
                const uint32_t sidx = c.cnt / kWave;
                seg_min[sidx] = (c.cnt % kWave) == 0 ? cd : fminf(seg_min[sidx], cd);
              }
            }
            c.cnt += 1;
          }
          // results
          const bool ok = (okm >> b) & 1ull;
          if (ok) top.insert(cd, cid, a.ef, lane);
          if (top.cnt) lowerBound = top.at_d(top.cnt - 1);
        }
      }
    }

    if (abandoned) {   // nothing is written for this query: the graph-sized launch answers it
      if (lane == 0) {
        const uint32_t slot = atomicAdd(a.redo_out, 1u);
        a.redo_out[1 + slot] = q;
      }
      uint32_t nxt = 0;
      if (lane == 0) nxt = atomicAdd(a.queue, 1u);
      qi = wstride + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
      continue;
    }
    st_eval += q_eval;
    st_hops += q_hops;

    // ---- trim to k, label, order by (distance,label) ---------------------------------------------
    const uint32_t kout = top.cnt < a.k ? top.cnt : a.k;
    float *od = a.out_dist + (size_t)q * a.k;
    uint64_t *ol = a.out_label + (size_t)q * a.k;
    if constexpr (kLdsList) {
      // the list is ascending by distance; equal distances form runs that are ordered by label
      for (uint32_t r = lane; r < a.k; r += kWave) {
        if (r < kout) {
          const float dr = top.d[r];
          const uint64_t lr = a.out_ids ? (uint64_t)top.id[r] : a.labels[top.id[r]];
          uint32_t lo = r, rank_in_run = 0;
          while (lo > 0 && top.d[lo - 1] == dr) --lo;
          for (uint32_t t = lo; t < kout && top.d[t] == dr; ++t) {
            if (t == r) continue;
            const uint64_t lt = a.out_ids ? (uint64_t)top.id[t] : a.labels[top.id[t]];
            rank_in_run += lt < lr ? 1u : 0u;
          }
          // (like the register path, only the first kout entries are ranked among themselves)
          const uint32_t pos = lo + rank_in_run;
          od[pos] = dr;
          ol[pos] = lr;
        } else {
          od[r] = __builtin_inff();
          ol[r] = kNoLabel;
        }
      }
    } else {
    uint64_t lab[kE > 0 ? kE : 1];
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      const uint32_t r = (uint32_t)e * kWave + lane;
      lab[e] = r < kout ? (a.out_ids ? (uint64_t)top.id[e] : a.labels[top.id[e]]) : kNoLabel;
    }
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      const uint32_t r = (uint32_t)e * kWave + lane;
      uint32_t rank = 0;
      for (uint32_t s = 0; s < kout; ++s) {
        float sd = 0.f;
        uint64_t sl = 0;
#pragma unroll
        for (int e2 = 0; e2 < kE; ++e2)
          if ((s >> 6) == (uint32_t)e2) { sd = readlane_f32(top.d[e2], s & 63); sl = readlane_u64(lab[e2], s & 63); }
        rank += dl_less(sd, sl, top.d[e], lab[e]) ? 1u : 0u;
      }
      if (r < kout) { od[rank] = top.d[e]; ol[rank] = lab[e]; }
      else if (r < a.k) { od[r] = __builtin_inff(); ol[r] = kNoLabel; }
    }
    }
    if (lane == 0) a.out_n[q] = kout;
    st_q += 1;
    uint32_t nxt = 0;
    if (lane == 0) nxt = atomicAdd(a.queue, 1u);
    qi = wstride + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
  }

  if (lane == 0 && a.stats && st_q) {
    atomicAdd(&a.stats[0], st_eval);
    atomicAdd(&a.stats[1], st_hops);
    atomicAdd(&a.stats[2], st_over);
    atomicAdd(&a.stats[3], st_q);
    if (a.redo_in) atomicAdd(&a.stats[4], st_q);   // ... of which answered by the graph-sized-frontier launch
    if (a.totals) {   // the index's running totals (device-buffer calls return before their statistics could be read back)
      atomicAdd(&a.totals[0], st_eval);
      atomicAdd(&a.totals[1], st_hops);
    }
  }
}

// 4 waves per SIMD (<= 128 VGPRs): a full batch is bound by gathers in flight, i.e. by resident waves
// (16 result slots per lane, 512 < ef <= 1024: 32 more registers than fit 128 -- three waves per SIMD instead of four,
// still twelve per CU where the LDS result list allows seven)
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, kE == 16 ? 3 : 4) void hnsw_search_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 8, false>(a);
}
// ... with the visited set as a hash table of ids (HnswSearchArgs::vis_hash_log2)
// (row pieces in flight per lane x waves per SIMD, same 10M graph and batch: 8 x 4 here; 12 x 3 and 24 x 2 +2.5 %,
// 16 x 3 the same, 16 x 4 and 12 x 4 -- spilling -- 2-4 % slower: the batch is bound by what the memory system does
// with this mix of accesses, not by one wave's chain of round trips)
// r03: three waves per SIMD (<= 168 VGPRs) with 12 pieces in flight for every list size -- the 128-register build of the
// f32 instantiations the 10M bench runs (kE = 2, 4) kept 20-24 registers in scratch, and scratch traffic is exactly the
// kind of access this kernel has no bandwidth to spare for.
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 3) void hnsw_search_hash_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, (kE <= 4 ? 12 : 8), false, 0, 1>(a);   // (8 and 16 slots per lane leave room for 8 pieces only)
}
// ... with the table in buckets whose fill counts live in LDS: no atomics on memory (HnswSearchArgs::vis_mode == 2)
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 3) void hnsw_search_bucket_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, (kE <= 4 ? 12 : 8), false, 0, 2>(a);
}
// ... with the set in LDS (vis_mode 3; result lists up to 256 entries): two waves per SIMD, each with twice the row pieces
// in flight -- what the 12 KB of the set cost in resident waves comes back as bytes per wave
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 2) void hnsw_search_ldsvis_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 24, false, 0, 3>(a);
}
// ... and the 32 KB set (result lists up to 512 entries): ONE wave per SIMD, a whole 768-element row in flight per lane
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 1) void hnsw_search_ldsvis_big_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 48, false, 0, 5>(a);
}
// searches with a filter or tombstones: the frontier lives in HBM (HnswSearchArgs::pool_g)
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 4) void hnsw_search_gpool_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 8, false, 1>(a);
}
// ... and the queries whose frontier outgrew that kernel's 64k entries, again, with a frontier sized by the graph
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 4) void hnsw_search_gpool2_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 8, false, 2>(a);
}
// a few queries cannot fill the device: each wave is alone with its memory latency, so it keeps six times as
// many row pieces in flight (a whole 768-d row per lane; one block per CU, registers instead of occupancy)
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 1) void hnsw_search_latency_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 48, true>(a);
}

// ... and the same for a few FILTERED queries (HBM frontier)
template <bool kL2, int kE, bool kBf16>
__global__ __launch_bounds__(256, 1) void hnsw_search_gpool_latency_kernel(HnswSearchArgs a) {
  hnsw_search_body<kL2, kE, kBf16, 48, true, 1>(a);
}

__global__ void scatter_u32_kernel(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride) {
  const uint64_t total = (uint64_t)n * stride;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t i = (uint32_t)(t / stride), w = (uint32_t)(t % stride);
    dst[(size_t)idx[i] * stride + w] = src[t];
  }
}

__global__ void gather_u32_kernel(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride) {
  const uint64_t total = (uint64_t)n * stride;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t i = (uint32_t)(t / stride), w = (uint32_t)(t % stride);
    dst[t] = src[(size_t)idx[i] * stride + w];
  }
}

// ---- host side ----------------------------------------------------------------------------------------
int hnsw_slots_per_lane(uint64_t ef) {
  if (ef <= 64) return 1;
  if (ef <= 128) return 2;
  if (ef <= 256) return 4;
  if (ef <= 512) return 8;
  if (ef <= 1024) return 16;                   // (LDS-frontier kernels; the others keep the list in LDS from 512 on)
  if (ef <= kHnswMaxEf) return kHnswLdsList;   // result list in LDS, one wave per block
  return 0;
}

// the result list of this launch lives in LDS (one wave per block) rather than in the lanes' registers
static bool hnsw_list_in_lds(const HnswSearchArgs &a) { return a.ef > 1024 || (a.ef > 512 && a.gpool_level != 0); }

static size_t hnsw_lds_per_wave(const HnswSearchArgs &a) {
  const bool lds_list = hnsw_list_in_lds(a);
  // HBM frontier: segment minima only (gpool_level 2: one per 64 segments)
  const size_t pool = a.gpool_level == 2 ? (size_t)(a.cand_cap / 8192) * 2
                      : a.gpool_level == 1 ? (size_t)(a.cand_cap / 128) * 2 : (size_t)a.cand_cap * 2;
  const size_t vis_cnt_words = a.vis_hash_log2 && a.vis_mode == 5 ? LdsVis<true>::kWords : a.vis_hash_log2 && a.vis_mode == 3 ? LdsVis<false>::kWords : a.vis_hash_log2 && a.vis_mode == 2 ? ((size_t)1 << a.vis_hash_log2) / (kVisBucket * 4) : 0;
  const size_t per_wave_f4 = (size_t)a.chunks * 4 + ((lds_list ? 2 * a.ef : 0) + pool + a.nbr_cap * 2 + vis_cnt_words + 3) / 4;
  return per_wave_f4 * 16;
}

// waves (queries) per block: 4, or fewer when the per-wave LDS (query + frontier pool, + result list for ef > 512)
// of four does not fit a CU -- rows beyond ~9000 dimensions, up to the FLAT limit, run 2 or 1 waves per block
static bool hnsw_latency_variant(const HnswSearchArgs &a);
int hnsw_waves_per_block(const HnswSearchArgs &a) {
  if (hnsw_list_in_lds(a)) return 1;
  // a batch too small to fill the device (the latency kernel): one query per block, so that 256 queries are on 256 CUs
  // and not four to a CU on 64 of them (1M x 768, ef = 128: 64 queries 1.42 -> 1.17 ms, 256 1.56 -> 1.32, 512 1.73 -> 1.56)
  // (the HBM-frontier kernels of a small filtered batch likewise)
  if (a.vis_hash_log2 == 0 && hnsw_latency_variant(a)) return 1;
  const size_t pw = hnsw_lds_per_wave(a);
  return 4 * pw <= 160 * 1024 ? 4 : 2 * pw <= 160 * 1024 ? 2 : 1;
}

size_t hnsw_lds_bytes(const HnswSearchArgs &a) { return hnsw_lds_per_wave(a) * (size_t)hnsw_waves_per_block(a); }

template <bool kL2, int kE, bool kBf16>
static const void *hnsw_fn(bool latency, int gpool, int hash) {
  if constexpr (kE >= 1 && kE <= 16) {   // (sixteen slots per lane: ef 513 .. 576, as far as eight waves fit a CU at 768 dimensions)
    if (hash == 3) return reinterpret_cast<const void *>(&hnsw_search_ldsvis_kernel<kL2, kE, kBf16>);
  }
  if constexpr (kE >= 1 && kE <= 8) {
    if (hash == 5) return reinterpret_cast<const void *>(&hnsw_search_ldsvis_big_kernel<kL2, kE, kBf16>);
  }
  if (hash == 3 || hash == 5) return nullptr;
  if (hash == 2) return reinterpret_cast<const void *>(&hnsw_search_bucket_kernel<kL2, kE, kBf16>);
  if (hash) return reinterpret_cast<const void *>(&hnsw_search_hash_kernel<kL2, kE, kBf16>);
  if constexpr (kE == 16) {   // (LDS-frontier kernels only)
    return gpool ? nullptr : reinterpret_cast<const void *>(&hnsw_search_kernel<kL2, kE, kBf16>);
  } else {
    if (gpool == 2) return reinterpret_cast<const void *>(&hnsw_search_gpool2_kernel<kL2, kE, kBf16>);
    if constexpr (kE >= 1 && kE <= 4) {
      if (gpool && latency) return reinterpret_cast<const void *>(&hnsw_search_gpool_latency_kernel<kL2, kE, kBf16>);
    }
    if (gpool) return reinterpret_cast<const void *>(&hnsw_search_gpool_kernel<kL2, kE, kBf16>);
    if constexpr (kE >= 1 && kE <= 4) {
      if (latency) return reinterpret_cast<const void *>(&hnsw_search_latency_kernel<kL2, kE, kBf16>);
    }
    return reinterpret_cast<const void *>(&hnsw_search_kernel<kL2, kE, kBf16>);
  }
}

template <int kE>
static const void *hnsw_pick_e(bool l2, bool bf16, bool latency, int gpool, int hash) {
  return l2 ? (bf16 ? hnsw_fn<true, kE, true>(latency, gpool, hash) : hnsw_fn<true, kE, false>(latency, gpool, hash))
            : (bf16 ? hnsw_fn<false, kE, true>(latency, gpool, hash) : hnsw_fn<false, kE, false>(latency, gpool, hash));
}

// a batch this small leaves most SIMDs without a wave: latency, not occupancy, is what counts
static bool hnsw_latency_variant(const HnswSearchArgs &a) {
  static const uint32_t max_nq = (uint32_t)VK_TUNE("VK_HNSW_LATENCY_NQ", 1024);   // (1M x 768, ef = 128, 1024 queries: 3.07 ms on the throughput kernel, 2.12 ms here)
  return a.nq <= max_nq;
}

static const void *hnsw_pick(const HnswSearchArgs &a, bool l2, bool bf16, int e) {
  const int hash = a.vis_hash_log2 == 0 ? 0 : a.vis_mode == 5 ? 5 : a.vis_mode == 3 ? 3 : a.vis_mode == 2 ? 2 : 1;
  if (hash && (a.gpool_level != 0 || a.redo_in != nullptr)) return nullptr;   // (LDS-frontier first launches only)
  const bool latency = !hash && hnsw_latency_variant(a);
  const int gpool = (int)a.gpool_level;
  switch (e) {
    case 1: return hnsw_pick_e<1>(l2, bf16, latency, gpool, hash);
    case 2: return hnsw_pick_e<2>(l2, bf16, latency, gpool, hash);
    case 4: return hnsw_pick_e<4>(l2, bf16, latency, gpool, hash);
    case 8: return hnsw_pick_e<8>(l2, bf16, latency, gpool, hash);
    case 16: return gpool ? hnsw_pick_e<0>(l2, bf16, latency, gpool, hash) : hnsw_pick_e<16>(l2, bf16, latency, gpool, hash);
    case kHnswLdsList: return hnsw_pick_e<0>(l2, bf16, latency, gpool, hash);
  }
  return nullptr;
}
bool hnsw_uses_latency_variant(const HnswSearchArgs &a) { return hnsw_latency_variant(a); }

hipError_t hnsw_max_blocks(const HnswSearchArgs &a, bool l2, bool bf16, int e, int *blocks) {
  const void *f = hnsw_pick(a, l2, bf16, e);
  if (!f) return hipErrorInvalidValue;
  const size_t lds = hnsw_lds_bytes(a);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    hipError_t er = ensure_max_lds(f);
    if (er != hipSuccess) return er;
  }
  int per_cu = 0;
  hipError_t er = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, 64 * hnsw_waves_per_block(a), lds);
  if (er != hipSuccess) return er;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (per_cu < 1) per_cu = 1;
  *blocks = per_cu * (cus > 0 ? cus : 256);
  return hipSuccess;
}

hipError_t launch_hnsw_search(const HnswSearchArgs &a, bool l2, bool bf16, int e, uint32_t blocks, hipStream_t s) {
  const void *f = hnsw_pick(a, l2, bf16, e);
  if (!f || blocks == 0) return hipErrorInvalidValue;
  const size_t lds = hnsw_lds_bytes(a);
  if (lds > 48 * 1024) {
    hipError_t er = ensure_max_lds(f);
    if (er != hipSuccess) return er;
  }
  HnswSearchArgs args = a;
  void *params[] = {&args};
  return hipLaunchKernel(f, dim3(blocks), dim3(64 * hnsw_waves_per_block(a)), params, lds, s);
}

hipError_t launch_scatter_u32(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride,
                              hipStream_t s) {
  if (n == 0) return hipSuccess;
  uint64_t total = (uint64_t)n * stride;
  uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(scatter_u32_kernel, dim3(blocks), dim3(256), 0, s, dst, src, idx, n, stride);
  return hipGetLastError();
}

hipError_t launch_gather_u32(uint32_t *dst, const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t stride,
                             hipStream_t s) {
  if (n == 0) return hipSuccess;
  uint64_t total = (uint64_t)n * stride;
  uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks), dim3(256), 0, s, dst, src, idx, n, stride);
  return hipGetLastError();
}

}  // namespace vk
