// index_common.cc -- host-side helpers every index type shares and that need nothing but the HIP runtime: the grow-only
// device / pinned buffers, the split of a batch into runs of queries that share a filter, the protobuf varint helpers of the
// index.proto headers.  (Kept apart from the kernels' translation units: the sharded index over fake shards -- the multi-device
// pre-flight, tests/helpers/san_sharded_main.cc -- links this file and no kernel.)
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "index.hpp"

namespace vk {

// ---- buffers -----------------------------------------------------------------------------------
Status DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return Status::Ok();
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  size_t want = std::max<size_t>(bytes, 4096);
  want = (want + 4095) & ~(size_t)4095;
  VK_HIP_TRY(hipMalloc(&p, want));
  cap = want;
  return Status::Ok();
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
Status PinBuf::ensure(size_t bytes) {
  if (bytes <= cap) return Status::Ok();
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  size_t want = std::max<size_t>(bytes, 4096);
  want = (want + 4095) & ~(size_t)4095;
  VK_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
  cap = want;
  return Status::Ok();
}
void PinBuf::release() {
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

// an index without per-query filters in its kernels: the batch split into runs of queries that share a bitmap
Status search_grouped_by_filter(Index *ix, const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
  const uint32_t dim = ix->params().dim;
  std::vector<uint8_t> done(rq.nq, 0);
  std::vector<float> Q, D;
  std::vector<uint64_t> L, N, idx;
  auto fs = [&](uint64_t q) { return rq.filter_tab ? rq.filter_tab[q] : nullptr; };
  auto ab = [&](uint64_t q) { return rq.allow_tab && !fs(q) ? rq.allow_tab[q] : nullptr; };
  auto nb = [&](uint64_t q) { return ab(q) ? rq.allow_nbits_tab[q] : 0; };
  for (uint64_t q0 = 0; q0 < rq.nq; ++q0) {
    if (done[q0]) continue;
    idx.clear();
    for (uint64_t q = q0; q < rq.nq; ++q)
      if (!done[q] && fs(q) == fs(q0) && ab(q) == ab(q0) && nb(q) == nb(q0)) { idx.push_back(q); done[q] = 1; }
    const uint64_t m = idx.size();
    Q.resize(m * dim);
    D.resize(m * rq.k);
    L.resize(m * rq.k);
    N.resize(m);
    for (uint64_t i = 0; i < m; ++i) memcpy(Q.data() + i * dim, rq.query_tab ? rq.query_tab[idx[i]] : rq.queries + idx[i] * dim, (size_t)dim * 4);
    SearchRequest g = rq;
    g.queries = Q.data();
    g.query_tab = nullptr;
    g.nq = m;
    g.allow_tab = nullptr;
    g.allow_nbits_tab = nullptr;
    g.filter_tab = nullptr;
    g.member_cancel = nullptr;
    g.filter = fs(q0);
    g.allow_bits = ab(q0);
    g.allow_nbits = nb(q0);
    VK_TRY(ix->search(g, D.data(), L.data(), N.data()));
    for (uint64_t i = 0; i < m; ++i) {
      out_n[idx[i]] = N[i];
      memcpy(out_dist + idx[i] * rq.k, D.data() + i * rq.k, (size_t)N[i] * 4);
      memcpy(out_label + idx[i] * rq.k, L.data() + i * rq.k, (size_t)N[i] * 8);
    }
  }
  return Status::Ok();
}

// ---- protobuf helpers --------------------------------------------------------------------------
static void pb_put_varint(std::string &s, uint64_t v) {
  while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
  s.push_back((char)v);
}
void pb_put_varint_field(std::string &s, uint32_t field, uint64_t v) {
  if (v == 0) return;  // proto3: default values are not serialised
  pb_put_varint(s, (uint64_t)field << 3);
  pb_put_varint(s, v);
}
void pb_put_double_field(std::string &s, uint32_t field, double v) {
  if (v == 0.0) return;
  pb_put_varint(s, ((uint64_t)field << 3) | 1);
  char b[8];
  memcpy(b, &v, 8);
  s.append(b, 8);
}
bool PbReader::next(uint32_t *field, uint32_t *wire, uint64_t *val) {
  auto varint = [&](uint64_t *o) {
    uint64_t v = 0;
    int shift = 0;
    while (p < end && shift < 64) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) { *o = v; return true; }
      shift += 7;
    }
    return false;
  };
  if (p >= end) return false;
  uint64_t key;
  if (!varint(&key)) return false;
  *field = (uint32_t)(key >> 3);
  *wire = (uint32_t)(key & 7);
  if (*wire == 0) return varint(val);
  if (*wire == 1) {
    if (end - p < 8) return false;
    memcpy(val, p, 8);
    p += 8;
    return true;
  }
  if (*wire == 5) {
    if (end - p < 4) return false;
    uint32_t v;
    memcpy(&v, p, 4);
    *val = v;
    p += 4;
    return true;
  }
  if (*wire == 2) {
    uint64_t n;
    if (!varint(&n) || (uint64_t)(end - p) < n) return false;
    p += n;
    *val = n;
    return true;
  }
  return false;
}

}  // namespace vk
