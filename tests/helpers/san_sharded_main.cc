// san_sharded_main.cc -- TEST INFRASTRUCTURE: the multi-device pre-flight of csrc/sharded_index.cc.
//
// The product's ShardedIndex (fan-out over one enqueue thread per device, queries broadcast by peer copy, per-shard lists
// gathered by peer copies or by an RCCL all-gather, (distance, label) merge, device-resident filters on every device) has only
// ever run on ONE physical GPU.  Here the REAL sharded_index.cc, index_common.cc and filter_set.cc are linked against a model
// of the HIP runtime with N virtual devices and asynchronous streams (hip_virtual.cc, which also models the five RCCL calls),
// with FAKE shards in place of FlatIndex / HnswIndex: a fake shard keeps its rows on the host and its "search kernel" is an
// exact scan enqueued on the stream it was given -- it reads the queries, the filters and writes its lists THROUGH THE DEVICE
// POINTERS the fan-out handed it, at the moment the stream reaches it.  What is checked, for device lists from 1 to 8
// devices with even and uneven shard counts per device and a serving device that is not device 0:
//   * every answer (host entry, device entry on a caller stream, host bitmaps, device-resident filters, one filter per
//     query, both gather modes, concurrent callers, contexts reused across batch shapes) equals the exact answer over all rows;
//   * the runtime model recorded NO violation: no copy or kernel touched memory of a device it must not touch, peer access
//     was enabled before every peer copy, every event was recorded on a stream of its own device, every rank joined every
//     collective with buffers of the right size on the right device;
//   * under -fsanitize=thread a missing event dependency is a data race, under -fsanitize=address an overrun of a gather
//     buffer is a heap overflow (tests/test_host_sanitizers.py builds both).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <random>
#include <thread>

#include "hip_virtual.hpp"
#include "index.hpp"
#include "shard_layout.hpp"

using vk::Status;

// ---- what the product's kernel translation units provide, as host models on the virtual streams ------------------------------
namespace vk {
constexpr uint64_t kNoLabel = ~0ull;   // device_common.hpp:28 (a header of device code)
thread_local const LoadObserver *g_load_observer = nullptr;

int flat_scan_slots_per_lane(uint64_t k) { return k <= 64 ? 1 : k <= 256 ? 4 : k <= 1024 ? 16 : 0; }   // flat_scan.hip

hipError_t launch_fill_empty(float *out_dist, uint64_t *out_label, uint32_t *out_n, uint32_t nq, uint32_t k, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  const size_t nk = (size_t)nq * k;
  hipv::launch(s, "fill_empty_kernel", {{out_dist, nk * 4, "out_dist"}, {out_label, nk * 8, "out_label"}, {out_n, out_n ? (size_t)nq * 4 : 0, "out_n"}},
               [=] {
                 for (size_t i = 0; i < nk; ++i) { out_dist[i] = INFINITY; out_label[i] = kNoLabel; }
                 if (out_n) for (uint32_t q = 0; q < nq; ++q) out_n[q] = 0;
               });
  return hipSuccess;
}

// merge_topk_kernel as the fan-out uses it: `parts` lists of per_part entries per query -> the k smallest by (distance, label)
hipError_t launch_merge_topk(const MergeArgs &m, int, uint64_t nq, hipStream_t s) {
  const size_t in_entries = (size_t)(m.parts - 1) * m.part_stride + (nq - 1) * m.q_stride + m.per_part;
  const uint32_t ld = m.out_ld ? m.out_ld : m.k;
  const MergeArgs a = m;
  hipv::launch(s, "merge_topk_kernel",
               {{m.in_dist, in_entries * 4, "in_dist"}, {m.in_label, in_entries * 8, "in_label"}, {m.out_dist, nq * ld * 4, "out_dist"},
                {m.out_label, nq * ld * 8, "out_label"}, {m.out_n, nq * 4, "out_n"}},
               [a, nq, ld] {
                 std::vector<std::pair<float, uint64_t>> all;
                 for (uint64_t q = 0; q < nq; ++q) {
                   all.clear();
                   for (uint32_t p = 0; p < a.parts; ++p)
                     for (uint32_t i = 0; i < a.per_part; ++i) {
                       const size_t at = (size_t)p * a.part_stride + q * a.q_stride + i;
                       if (a.in_label[at] != kNoLabel) all.emplace_back(a.in_dist[at], a.in_label[at]);
                     }
                   std::sort(all.begin(), all.end());
                   const uint32_t n = (uint32_t)std::min<size_t>(all.size(), a.k);
                   for (uint32_t i = 0; i < ld; ++i) {
                     a.out_dist[q * ld + i] = i < n ? all[i].first : INFINITY;
                     a.out_label[q * ld + i] = i < n ? all[i].second : kNoLabel;
                   }
                   a.out_n[q] = n;
                 }
               });
  return hipSuccess;
}

// filter_build.hip
uint32_t filter_set_ids_blocks(uint64_t n) { return n ? 1 : 0; }
uint32_t filter_set_runs_blocks(uint64_t n) { return n ? 1 : 0; }
uint32_t filter_combine_blocks(uint64_t) { return 1; }
hipError_t launch_filter_set_ids(uint64_t *bits, uint64_t nbits, const uint64_t *d_ids, uint64_t n, unsigned long long *d_partial, hipStream_t s) {
  hipv::launch(s, "filter_set_ids_kernel", {{bits, (size_t)((nbits + 63) / 64 + 1) * 8, "bits"}, {d_ids, (size_t)n * 8, "ids"}, {d_partial, 8, "partial"}}, [=] {
    unsigned long long on = 0;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t id = d_ids[i];
      if (id >= nbits) continue;
      const uint64_t w = bits[id >> 6], b = 1ull << (id & 63);
      if (!(w & b)) { bits[id >> 6] = w | b; ++on; }
    }
    d_partial[0] = on;
  });
  return hipSuccess;
}
hipError_t launch_filter_set_runs(uint64_t *bits, uint64_t nbits, const uint64_t *d_runs, uint64_t n_runs, unsigned long long *d_partial, hipStream_t s) {
  hipv::launch(s, "filter_set_runs_kernel", {{bits, (size_t)((nbits + 63) / 64 + 1) * 8, "bits"}, {d_runs, (size_t)n_runs * 16, "runs"}, {d_partial, 8, "partial"}}, [=] {
    unsigned long long on = 0;
    for (uint64_t r = 0; r < n_runs; ++r)
      for (uint64_t id = d_runs[2 * r]; id <= d_runs[2 * r + 1] && id < nbits; ++id) {
        const uint64_t w = bits[id >> 6], b = 1ull << (id & 63);
        if (!(w & b)) { bits[id >> 6] = w | b; ++on; }
      }
    d_partial[0] = on;
  });
  return hipSuccess;
}
hipError_t launch_filter_popcount(const uint64_t *bits, uint64_t words, unsigned long long *d_out, hipStream_t s) {
  hipv::launch(s, "filter_popcount_kernel", {{bits, (size_t)words * 8, "bits"}, {d_out, 8, "count"}}, [=] {
    unsigned long long c = 0;
    for (uint64_t i = 0; i < words; ++i) c += (unsigned long long)__builtin_popcountll(bits[i]);
    *d_out += c;
  });
  return hipSuccess;
}
hipError_t launch_filter_combine(uint64_t *dst, const uint64_t *a, const uint64_t *b, uint64_t words, uint32_t op, unsigned long long *d_partial, hipStream_t s) {
  hipv::launch(s, "filter_combine_kernel", {{dst, (size_t)(words + 1) * 8, "dst"}, {a, (size_t)words * 8, "a"}, {b, (size_t)words * 8, "b"}, {d_partial, 8, "partial"}}, [=] {
    unsigned long long c = 0;
    for (uint64_t i = 0; i < words; ++i) {
      dst[i] = op == 0 ? (a[i] & b[i]) : op == 1 ? (a[i] | b[i]) : (a[i] & ~b[i]);
      c += (unsigned long long)__builtin_popcountll(dst[i]);
    }
    dst[words] = 0;
    d_partial[0] = c;
  });
  return hipSuccess;
}
hipError_t launch_filter_combine_batch(const uint64_t *d_items, uint32_t n, uint64_t words, unsigned long long *d_counts, hipStream_t s) {
  std::vector<hipv::Access> touched = {{d_items, (size_t)n * 32, "items"}, {d_counts, (size_t)n * 8, "counts"}};
  hipv::launch(s, "filter_combine_batch_kernel", touched, [=] {
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t *dst = reinterpret_cast<uint64_t *>(d_items[4 * i]);
      const uint64_t *a = reinterpret_cast<const uint64_t *>(d_items[4 * i + 1]), *b = reinterpret_cast<const uint64_t *>(d_items[4 * i + 2]);
      const uint64_t op = d_items[4 * i + 3];
      // (the operands named inside the table are checked here, where they are known: each must lie on this stream's device)
      const int dev = hipv::stream_device(s);
      if (hipv::memory_device(dst, (words + 1) * 8) != dev || hipv::memory_device(a, words * 8) != dev || hipv::memory_device(b, words * 8) != dev) {
        fprintf(stderr, "VIOLATION filter_combine_batch_kernel: item %u names memory of another device\n", i);
        abort();
      }
      unsigned long long c = 0;
      for (uint64_t w = 0; w < words; ++w) {
        dst[w] = op == 0 ? (a[w] & b[w]) : op == 1 ? (a[w] | b[w]) : (a[w] & ~b[w]);
        c += (unsigned long long)__builtin_popcountll(dst[w]);
      }
      dst[words] = 0;
      d_counts[i] += c;
    }
  });
  return hipSuccess;
}
}  // namespace vk

// ---- a fake shard ----------------------------------------------------------------------------------------------------------
namespace {
std::atomic<uint64_t> g_bad{0};
#define CHECK(cond, ...)                                                     \
  do {                                                                       \
    if (!(cond)) {                                                           \
      g_bad.fetch_add(1);                                                    \
      fprintf(stderr, "CHECK failed %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
      fprintf(stderr, __VA_ARGS__);                                          \
      fprintf(stderr, "\n");                                                 \
    }                                                                        \
  } while (0)

float exact_distance(bool l2, const float *a, const float *b, uint32_t dim) {
  float acc = 0.f;
  for (uint32_t i = 0; i < dim; ++i) acc += l2 ? (a[i] - b[i]) * (a[i] - b[i]) : a[i] * b[i];
  return l2 ? acc : 1.0f - acc;
}
bool bit(const uint64_t *bits, uint64_t nbits, uint64_t label) { return label < nbits && ((bits[label >> 6] >> (label & 63)) & 1u); }

using Rows = std::map<uint64_t, std::vector<float>>;
// exact top-k by (distance, label) over `rows`, written like a kernel writes it: k entries per query, padded
void scan_rows(const Rows &rows, bool l2, uint32_t dim, const float *q, uint64_t k, const uint64_t *allow, uint64_t nbits, float *od, uint64_t *ol,
               uint32_t *on) {
  std::vector<std::pair<float, uint64_t>> all;
  for (const auto &kv : rows)
    if (!allow || bit(allow, nbits, kv.first)) all.emplace_back(exact_distance(l2, q, kv.second.data(), dim), kv.first);
  std::sort(all.begin(), all.end());
  const uint64_t n = std::min<uint64_t>(all.size(), k);
  for (uint64_t i = 0; i < k; ++i) { od[i] = i < n ? all[i].first : INFINITY; ol[i] = i < n ? all[i].second : vk::kNoLabel; }
  *on = (uint32_t)n;
}

class FakeShard final : public vk::Index {
 public:
  FakeShard(const vk_index_params &p, int device) : Index(p), device_(device), cap_(p.initial_cap) {}
  Status add(uint64_t label, const float *row) override {
    std::lock_guard<std::mutex> lk(mu_);
    if (!rows_.count(label) && rows_.size() >= cap_) return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
    rows_[label].assign(row, row + params_.dim);
    return Status::Ok();
  }
  Status add_batch(const uint64_t *labels, const float *rows, uint64_t n) override {
    for (uint64_t i = 0; i < n; ++i) VK_TRY(add(labels ? labels[i] : i, rows + i * params_.dim));
    return Status::Ok();
  }
  Status remove(uint64_t label) override {
    std::lock_guard<std::mutex> lk(mu_);
    return rows_.erase(label) || params_.algo == VK_ALGO_FLAT ? Status::Ok() : Status::Err(VK_ERR_NOT_FOUND, "Label not found");
  }
  Status resize(uint64_t n) override { std::lock_guard<std::mutex> lk(mu_); cap_ = n; return Status::Ok(); }
  Status set_ef(uint32_t) override { return Status::Ok(); }
  Status flush() override { return Status::Ok(); }
  void filter_devices(std::vector<int> *out) const override { *out = {device_}; }

  Status search(const vk::SearchRequest &rq, float *od, uint64_t *ol, uint64_t *on) override {
    std::lock_guard<std::mutex> lk(mu_);
    std::vector<uint64_t> host_bits;
    const uint64_t *allow = rq.allow_bits;
    uint64_t nbits = rq.allow_nbits;
    if (rq.filter) {
      host_bits.resize(rq.filter->words() + 1);
      VK_TRY(rq.filter->read(host_bits.data(), host_bits.size()));
      allow = host_bits.data();
      nbits = rq.filter->nbits();
    }
    for (uint64_t q = 0; q < rq.nq; ++q) {
      uint32_t n;
      scan_rows(rows_, l2(), params_.dim, rq.queries + q * params_.dim, rq.k, allow, nbits, od + q * rq.k, ol + q * rq.k, &n);
      on[q] = n;
    }
    return Status::Ok();
  }

  // the part under test: device pointers, the caller's stream, no host sync
  Status search_device(const vk::SearchRequest &rq, float *d_od, uint64_t *d_ol, uint32_t *d_on, hipStream_t stream) override {
    int cur = -1;
    (void)hipGetDevice(&cur);
    CHECK(stream != nullptr, "a shard of a sharded index is always given its lane's stream");
    CHECK(hipv::stream_device(stream) == device_, "shard on device %d was handed a stream of device %d", device_, hipv::stream_device(stream));
    (void)hipSetDevice(device_);   // (as the real indexes do)
    const uint64_t nq = rq.nq, k = rq.k, dim = params_.dim;
    std::vector<hipv::Access> touched = {{rq.queries, nq * dim * 4, "queries"}, {d_od, nq * k * 4, "out_dist"}, {d_ol, nq * k * 8, "out_label"},
                                         {d_on, nq * 4, "out_n"}};
    const uint64_t *allow = rq.allow_bits;
    const uint64_t nbits = rq.allow_nbits;
    if (allow) touched.push_back({allow, (size_t)((nbits + 63) / 64) * 8, "allow_bits"});
    std::vector<const uint64_t *> tab;       // one filter per query: pointed at where the copies lie on THIS device
    std::vector<uint64_t> tab_nbits;
    if (rq.filter_tab) {
      for (uint64_t q = 0; q < nq; ++q) {
        const vk::FilterSet *f = rq.filter_tab[q];
        const uint64_t *b = f ? f->bits_on(device_) : nullptr;
        if (f && !b) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's device");
        tab.push_back(b);
        tab_nbits.push_back(f ? f->nbits() : 0);
        if (b) touched.push_back({b, (size_t)f->words() * 8, "filter_tab[q]"});
      }
    }
    if (rq.cancel_word) touched.push_back({rq.cancel_word, 4, "cancel_word"});
    CHECK(rq.filter == nullptr && rq.cancel_flag == nullptr, "the fan-out resolves filter handles and host flags before it calls a shard");
    const float *dq = rq.queries;
    hipv::launch(stream, "fake_shard_search", std::move(touched), [this, dq, nq, k, dim, allow, nbits, tab, tab_nbits, d_od, d_ol, d_on] {
      std::lock_guard<std::mutex> lk(mu_);
      for (uint64_t q = 0; q < nq; ++q) {
        const uint64_t *a = !tab.empty() ? tab[q] : allow;
        const uint64_t nb = !tab.empty() ? tab_nbits[q] : nbits;
        scan_rows(rows_, l2(), (uint32_t)dim, dq + q * dim, k, a, nb, d_od + q * k, d_ol + q * k, d_on + q);
      }
    });
    return Status::Ok();
  }
  Status label_distances(const float *query, const uint64_t *labels, uint64_t n, float *out, uint8_t *found) override {
    std::lock_guard<std::mutex> lk(mu_);
    for (uint64_t i = 0; i < n; ++i) {
      auto it = rows_.find(labels[i]);
      found[i] = it != rows_.end();
      if (found[i]) out[i] = exact_distance(l2(), query, it->second.data(), params_.dim);
    }
    return Status::Ok();
  }
  Status distance(uint64_t label, const float *query, float *out) override {
    uint8_t f = 0;
    VK_TRY(label_distances(query, &label, 1, out, &f));
    return f ? Status::Ok() : Status::Err(VK_ERR_NOT_FOUND, "Couldn't find internal id");
  }
  Status get_row(uint64_t label, float *out) override {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = rows_.find(label);
    if (it == rows_.end()) return Status::Err(VK_ERR_NOT_FOUND, "label not found");
    memcpy(out, it->second.data(), params_.dim * 4);
    return Status::Ok();
  }
  Status contains(uint64_t label, bool *found) override { std::lock_guard<std::mutex> lk(mu_); *found = rows_.count(label) != 0; return Status::Ok(); }
  Status stats(vk_index_stats *out) override {
    std::lock_guard<std::mutex> lk(mu_);
    memset(out, 0, sizeof *out);
    out->count = rows_.size();
    out->capacity = cap_;
    out->max_level = -1;
    return Status::Ok();
  }
  Status device_rows(uint64_t, void **, uint64_t *) override { return Status::Err(VK_ERR_INVALID, "fake shard"); }
  Status commit_device_rows(uint64_t, const uint64_t *) override { return Status::Err(VK_ERR_INVALID, "fake shard"); }
  // FLAT SaveIndex layout (bruteforce.h:147-207): header, then [row | label] per element
  Status save(vk_write_chunk_fn fn, void *user) override {
    std::lock_guard<std::mutex> lk(mu_);
    std::string hdr;
    vk::pb_put_varint_field(hdr, 1, cap_);
    vk::pb_put_varint_field(hdr, 2, (uint64_t)params_.dim * 4 + 8);
    vk::pb_put_varint_field(hdr, 3, rows_.size());
    if (fn(user, hdr.data(), hdr.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
    std::vector<char> e((size_t)params_.dim * 4 + 8);
    for (const auto &kv : rows_) {
      memcpy(e.data(), kv.second.data(), (size_t)params_.dim * 4);
      memcpy(e.data() + (size_t)params_.dim * 4, &kv.first, 8);
      if (fn(user, e.data(), e.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
    }
    return Status::Ok();
  }
  int device() const { return device_; }

 private:
  int device_;
  std::mutex mu_;
  Rows rows_;
  uint64_t cap_;
};
}  // namespace

namespace vk {
static Status make_fake(const vk_index_params &p, std::unique_ptr<Index> *out) {
  int n = 0;
  (void)hipGetDeviceCount(&n);
  if (p.device_id < 0 || p.device_id >= n) return Status::Err(VK_ERR_INVALID, "fake shard: device out of range");
  *out = std::make_unique<FakeShard>(p, p.device_id);
  return Status::Ok();
}
Status create_flat(const vk_index_params &p, std::unique_ptr<Index> *out) { return make_fake(p, out); }
Status create_hnsw(const vk_index_params &p, std::unique_ptr<Index> *out) { return make_fake(p, out); }
Status load_flat(const vk_index_params &, vk_read_chunk_fn, void *, std::unique_ptr<Index> *) { return Status::Err(VK_ERR_INVALID, "fake shard"); }
Status load_hnsw(const vk_index_params &, vk_read_chunk_fn, void *, std::unique_ptr<Index> *) { return Status::Err(VK_ERR_INVALID, "fake shard"); }
}  // namespace vk

// ---- the scenarios -----------------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kDim = 8;

struct World {
  std::vector<int> devices;
  bool hnsw;
  std::unique_ptr<vk::Index> ix;
  Rows rows;                       // what the index holds, for the exact answer
  std::mt19937_64 rng{12345};
  std::vector<float> random_vec(size_t n) {
    std::normal_distribution<float> d;
    std::vector<float> v(n);
    for (float &x : v) x = d(rng);
    return v;
  }
};

std::string dev_list(const std::vector<int> &d) {
  std::string s = "[";
  for (size_t i = 0; i < d.size(); ++i) s += (i ? "," : "") + std::to_string(d[i]);
  return s + "]";
}

void expect_answer(World &w, const char *what, const float *Q, uint64_t nq, uint64_t k, const uint64_t *allow, uint64_t nbits,
                   const std::vector<const uint64_t *> *per_q, const float *D, const uint64_t *L, const uint64_t *N32or64, bool n_is_u32) {
  std::vector<float> ed(k);
  std::vector<uint64_t> el(k);
  for (uint64_t q = 0; q < nq; ++q) {
    uint32_t en;
    scan_rows(w.rows, true, kDim, Q + q * kDim, k, per_q ? (*per_q)[q] : allow, nbits, ed.data(), el.data(), &en);
    const uint64_t n = n_is_u32 ? reinterpret_cast<const uint32_t *>(N32or64)[q] : N32or64[q];
    CHECK(n == en, "%s %s: query %llu returned %llu entries, exact answer has %u", dev_list(w.devices).c_str(), what, (unsigned long long)q,
          (unsigned long long)n, en);
    for (uint64_t i = 0; i < std::min<uint64_t>(n, en); ++i)
      CHECK(L[q * k + i] == el[i] && memcmp(&D[q * k + i], &ed[i], 4) == 0, "%s %s: query %llu entry %llu is (%g, %llu), exact (%g, %llu)",
            dev_list(w.devices).c_str(), what, (unsigned long long)q, (unsigned long long)i, D[q * k + i], (unsigned long long)L[q * k + i], ed[i],
            (unsigned long long)el[i]);
  }
}

void host_search(World &w, const char *what, uint64_t nq, uint64_t k, const uint64_t *allow, uint64_t nbits, const vk::FilterSet *filter,
                 const std::vector<uint64_t> *filter_bits) {
  std::vector<float> Q = w.random_vec(nq * kDim), D(nq * k);
  std::vector<uint64_t> L(nq * k), N(nq);
  vk::SearchRequest rq;
  rq.queries = Q.data();
  rq.nq = nq;
  rq.k = k;
  rq.allow_bits = allow;
  rq.allow_nbits = nbits;
  rq.filter = filter;
  Status st = w.ix->search(rq, D.data(), L.data(), N.data());
  CHECK(st.ok(), "%s %s: %s", dev_list(w.devices).c_str(), what, st.msg.c_str());
  if (!st.ok()) return;
  const uint64_t *bits = filter ? filter_bits->data() : allow;
  expect_answer(w, what, Q.data(), nq, k, bits, filter ? filter->nbits() : nbits, nullptr, D.data(), L.data(), N.data(), false);
}

// vk_index_search_batch_device: queries and answers on the serving device, the caller's stream, two calls in flight
void device_search(World &w, const char *what, uint64_t nq, uint64_t k) {
  const int dev0 = w.devices[0];
  (void)hipSetDevice(dev0);
  struct Call { hipStream_t s; float *dq; float *dd; uint64_t *dl; uint32_t *dn; std::vector<float> Q, D; std::vector<uint64_t> L; std::vector<uint32_t> N; };
  Call c[2];
  for (Call &x : c) {
    (void)hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking);
    x.Q = w.random_vec(nq * kDim);
    x.D.resize(nq * k);
    x.L.resize(nq * k);
    x.N.resize(nq);
    (void)hipMalloc(reinterpret_cast<void **>(&x.dq), nq * kDim * 4);
    (void)hipMalloc(reinterpret_cast<void **>(&x.dd), nq * k * 4);
    (void)hipMalloc(reinterpret_cast<void **>(&x.dl), nq * k * 8);
    (void)hipMalloc(reinterpret_cast<void **>(&x.dn), nq * 4);
    (void)hipMemcpyAsync(x.dq, x.Q.data(), nq * kDim * 4, hipMemcpyHostToDevice, x.s);
  }
  for (Call &x : c) {
    vk::SearchRequest rq;
    rq.queries = x.dq;
    rq.nq = nq;
    rq.k = k;
    Status st = w.ix->search_device(rq, x.dd, x.dl, x.dn, x.s);
    CHECK(st.ok(), "%s %s: %s", dev_list(w.devices).c_str(), what, st.msg.c_str());
    (void)hipSetDevice(dev0);
  }
  for (Call &x : c) {
    (void)hipMemcpyAsync(x.D.data(), x.dd, nq * k * 4, hipMemcpyDeviceToHost, x.s);
    (void)hipMemcpyAsync(x.L.data(), x.dl, nq * k * 8, hipMemcpyDeviceToHost, x.s);
    (void)hipMemcpyAsync(x.N.data(), x.dn, nq * 4, hipMemcpyDeviceToHost, x.s);
    (void)hipStreamSynchronize(x.s);
    expect_answer(w, what, x.Q.data(), nq, k, nullptr, 0, nullptr, x.D.data(), x.L.data(), reinterpret_cast<const uint64_t *>(x.N.data()), true);
  }
  for (Call &x : c) {
    (void)hipSetDevice(dev0);
    (void)hipFree(x.dq); (void)hipFree(x.dd); (void)hipFree(x.dl); (void)hipFree(x.dn);
    (void)hipStreamDestroy(x.s);
  }
}

void report_violations(const World &w, const char *stage) {
  for (const std::string &v : hipv::take_violations()) {
    g_bad.fetch_add(1);
    fprintf(stderr, "VIOLATION %s (%s): %s\n", dev_list(w.devices).c_str(), stage, v.c_str());
  }
}

void run_world(const std::vector<int> &devices, bool hnsw, bool threads) {
  World w;
  w.devices = devices;
  w.hnsw = hnsw;
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.algo = hnsw ? VK_ALGO_HNSW : VK_ALGO_FLAT;
  p.metric = VK_METRIC_L2;
  p.dim = kDim;
  p.initial_cap = 4000;
  p.m = 16;
  p.ef_construction = 100;
  p.ef_runtime = 64;
  p.n_shards = (uint32_t)devices.size();
  for (size_t i = 0; i < devices.size(); ++i) p.shard_devices[i] = devices[i];
  setenv("VK_SHARD_THREADS", threads ? "1" : "0", 1);
  Status st = vk::create_sharded(p, &w.ix);
  CHECK(st.ok(), "%s create: %s", dev_list(devices).c_str(), st.msg.c_str());
  if (!st.ok()) return;
  CHECK(w.ix->shard_count() == devices.size(), "shard count");
  std::vector<int> fdev;
  w.ix->filter_devices(&fdev);
  CHECK(fdev == devices, "filter_devices");

  // ---- ingest: a bulk, single adds, an update, removes
  const uint64_t n_bulk = 1200, n_single = 300;
  std::vector<uint64_t> labels;
  for (uint64_t i = 0; i < n_bulk + n_single; ++i) labels.push_back(i * 3 + 1);
  std::vector<float> X = w.random_vec((n_bulk + n_single) * kDim);
  st = w.ix->add_batch(labels.data(), X.data(), n_bulk);
  CHECK(st.ok(), "add_batch: %s", st.msg.c_str());
  for (uint64_t i = n_bulk; i < n_bulk + n_single; ++i) CHECK(w.ix->add(labels[i], X.data() + i * kDim).ok(), "add");
  for (uint64_t i = 0; i < n_bulk + n_single; ++i) w.rows[labels[i]].assign(X.begin() + i * kDim, X.begin() + (i + 1) * kDim);
  {   // the same label again: an update on the shard that holds it
    std::vector<float> v = w.random_vec(kDim);
    CHECK(w.ix->add(labels[7], v.data()).ok(), "update");
    w.rows[labels[7]] = v;
  }
  for (uint64_t i = 0; i < 90; i += 3) {
    CHECK(w.ix->remove(labels[i]).ok(), "remove");
    w.rows.erase(labels[i]);
  }
  vk_index_stats s0;
  CHECK(w.ix->stats(&s0).ok() && s0.count == w.rows.size(), "stats.count %llu vs %zu", (unsigned long long)s0.count, w.rows.size());
  {   // contiguous runs, evened out: no shard holds more than its share + the single adds
    uint64_t mx = 0, mn = ~0ull;
    for (uint32_t s = 0; s < devices.size(); ++s) {
      vk_index_stats t;
      CHECK(w.ix->shard_stats(s, &t).ok(), "shard_stats");
      mx = std::max(mx, t.count);
      mn = std::min(mn, t.count);
    }
    CHECK(mx - mn <= 40, "rows are dealt evenly: %llu .. %llu", (unsigned long long)mn, (unsigned long long)mx);
  }
  report_violations(w, "ingest");

  const uint64_t nbits = 3 * (n_bulk + n_single) + 64;
  std::vector<uint64_t> allow((nbits + 63) / 64 + 1, 0);
  for (uint64_t l = 0; l < nbits; ++l)
    if ((l * 2654435761u >> 7) % 10 < 3) allow[l >> 6] |= 1ull << (l & 63);

  for (int gather = 0; gather < 2; ++gather) {
    CHECK(w.ix->set_option("shard-gather", gather).ok(), "set_option");
    const char *gm = gather ? "all-gather" : "peer copies";
    const uint64_t peer0 = hipv::peer_copies(), col0 = hipv::collectives();
    vk_index_stats a0;
    (void)w.ix->stats(&a0);
    const std::pair<uint64_t, uint64_t> shapes[] = {{1, 1}, {5, 10}, {64, 10}, {33, 3}, {5, 10}, {2, 100}};
    for (auto sh : shapes) {
      host_search(w, gm, sh.first, sh.second, nullptr, 0, nullptr, nullptr);
      host_search(w, gm, sh.first, sh.second, allow.data(), nbits, nullptr, nullptr);
    }
    device_search(w, gm, 16, 10);
    device_search(w, gm, 3, 5);
    // device-resident filters: built once on every device of the index, handed to the shards where they lie
    std::shared_ptr<vk::FilterSet> fa, fb, fc;
    std::vector<uint64_t> ids, runs = {30, 400, 1000, 1003, 4000, 4400};
    for (uint64_t l = 1; l < nbits; l += 7) ids.push_back(l);
    st = vk::FilterSet::build(devices, nbits, ids.data(), ids.size(), nullptr, 0, nullptr, &fa);
    CHECK(st.ok(), "filter build: %s", st.msg.c_str());
    st = vk::FilterSet::build(devices, nbits, nullptr, 0, runs.data(), runs.size() / 2, nullptr, &fb);
    CHECK(st.ok(), "filter build (runs): %s", st.msg.c_str());
    if (fa && fb) {
      st = vk::FilterSet::combine(*fa, *fb, 1, &fc);
      CHECK(st.ok(), "filter combine: %s", st.msg.c_str());
    }
    if (fa && fb && fc) {
      std::vector<uint64_t> ba(fa->words() + 1), bb(fb->words() + 1), bc(fc->words() + 1);
      CHECK(fa->read(ba.data(), ba.size()).ok() && fb->read(bb.data(), bb.size()).ok() && fc->read(bc.data(), bc.size()).ok(), "filter read");
      uint64_t cnt = 0;
      for (uint64_t i = 0; i < fc->words(); ++i) { CHECK(bc[i] == (ba[i] | bb[i]), "combine word %llu", (unsigned long long)i); cnt += __builtin_popcountll(bc[i]); }
      CHECK(fc->allowed() == cnt && fa->allowed() == ids.size(), "allowed() counts");
      for (int d : devices) CHECK(fc->bits_on(d) != nullptr && hipv::memory_device(fc->bits_on(d), fc->words() * 8) == d, "a copy on device %d", d);
      {   // a batch of combinations in one launch per device: the same bitmaps and counts as one by one
        const vk::FilterSet *as[] = {fa.get(), fb.get(), fa.get()}, *bs[] = {fb.get(), fa.get(), fb.get()};
        const uint32_t ops[] = {1, 0, 2};
        std::vector<std::shared_ptr<vk::FilterSet>> many;
        st = vk::FilterSet::combine_batch(as, bs, ops, 3, &many);
        CHECK(st.ok() && many.size() == 3, "combine_batch: %s", st.msg.c_str());
        if (st.ok()) {
          std::vector<uint64_t> got(fc->words() + 1);
          CHECK(many[0]->read(got.data(), got.size()).ok() && got == bc && many[0]->allowed() == fc->allowed(), "combine_batch[0] = a | b");
          CHECK(many[1]->read(got.data(), got.size()).ok(), "read");
          for (uint64_t i = 0; i < fc->words(); ++i) CHECK(got[i] == (ba[i] & bb[i]), "combine_batch[1] = b & a, word %llu", (unsigned long long)i);
          for (int d : devices) CHECK(many[2]->bits_on(d) != nullptr && hipv::memory_device(many[2]->bits_on(d), fc->words() * 8) == d, "a copy on device %d", d);
        }
      }
      host_search(w, gm, 9, 10, nullptr, 0, fa.get(), &ba);
      host_search(w, gm, 40, 4, nullptr, 0, fc.get(), &bc);
      // one filter per query (the dispatcher's batches of hybrid queries): HNSW hands the table to the shards, FLAT runs per filter
      const uint64_t nq = 12, k = 6;
      std::vector<float> Q = w.random_vec(nq * kDim), D(nq * k);
      std::vector<uint64_t> L(nq * k), N(nq);
      std::vector<const vk::FilterSet *> tab(nq);
      std::vector<const uint64_t *> exact_bits(nq);
      for (uint64_t q = 0; q < nq; ++q) {
        tab[q] = q % 3 == 0 ? fa.get() : q % 3 == 1 ? fc.get() : nullptr;
        exact_bits[q] = q % 3 == 0 ? ba.data() : q % 3 == 1 ? bc.data() : nullptr;
      }
      vk::SearchRequest rq;
      rq.queries = Q.data();
      rq.nq = nq;
      rq.k = k;
      rq.filter_tab = tab.data();
      st = w.ix->search(rq, D.data(), L.data(), N.data());
      CHECK(st.ok(), "per-query filters: %s", st.msg.c_str());
      if (st.ok()) expect_answer(w, "per-query filters", Q.data(), nq, k, nullptr, nbits, &exact_bits, D.data(), L.data(), N.data(), false);
    }
    // concurrent callers: contexts from the pool, the collectives of one communicator set issued by one thread at a time
    {
      std::vector<std::thread> th;
      for (int t = 0; t < 4; ++t)
        th.emplace_back([&w, gm, t] {
          World local;   // (own rng; the index and the rows are shared, read-only here)
          local.devices = w.devices;
          local.rng.seed(1000 + t);
          for (int i = 0; i < 6; ++i) {
            const uint64_t nq = 1 + (uint64_t)(t * 5 + i) % 17, k = 1 + (uint64_t)(i * 3) % 12;
            std::vector<float> Q = local.random_vec(nq * kDim), D(nq * k);
            std::vector<uint64_t> L(nq * k), N(nq);
            vk::SearchRequest rq;
            rq.queries = Q.data();
            rq.nq = nq;
            rq.k = k;
            Status st = w.ix->search(rq, D.data(), L.data(), N.data());
            CHECK(st.ok(), "concurrent search: %s", st.msg.c_str());
            if (st.ok()) expect_answer(w, gm, Q.data(), nq, k, nullptr, 0, nullptr, D.data(), L.data(), N.data(), false);
          }
        });
      for (auto &t : th) t.join();
    }
    // a cancel flag that is never raised rides along (the relay word lives in pinned memory every device's kernels may read)
    {
      volatile int flag = 0;
      std::vector<float> Q = w.random_vec(4 * kDim), D(40);
      std::vector<uint64_t> L(40), N(4);
      vk::SearchRequest rq;
      rq.queries = Q.data();
      rq.nq = 4;
      rq.k = 10;
      rq.cancel_flag = &flag;
      st = w.ix->search(rq, D.data(), L.data(), N.data());
      CHECK(st.ok(), "search with a cancel flag: %s", st.msg.c_str());
      if (st.ok()) expect_answer(w, "cancel flag", Q.data(), 4, 10, nullptr, 0, nullptr, D.data(), L.data(), N.data(), false);
    }
    vk_index_stats a1;
    (void)w.ix->stats(&a1);
    const vk::ShardLayout lay = vk::ShardLayout::from_devices(devices);
    CHECK(a1.fanout_calls > a0.fanout_calls, "fan-outs counted");
    if (gather) {
      CHECK(a1.rccl_gathers > a0.rccl_gathers, "the all-gather ran (%llu -> %llu)", (unsigned long long)a0.rccl_gathers, (unsigned long long)a1.rccl_gathers);
      CHECK(hipv::collectives() - col0 == 2 * (a1.rccl_gathers - a0.rccl_gathers), "two collectives (distances, labels) per gather");
    } else {
      CHECK(a1.rccl_gathers == a0.rccl_gathers && hipv::collectives() == col0, "no collective in peer-copy mode");
    }
    if (lay.G() > 1) CHECK(hipv::peer_copies() > peer0, "queries were broadcast to the other devices");
    else CHECK(hipv::peer_copies() == peer0, "one device: nothing crosses a link");
    report_violations(w, gm);
  }
  // FLAT: one stream in the reference's layout, loadable with any shard count
  if (!hnsw) {
    struct Sink { std::vector<std::string> chunks; } sink;
    st = w.ix->save([](void *u, const void *d, uint64_t n) -> int { static_cast<Sink *>(u)->chunks.emplace_back(static_cast<const char *>(d), n); return 0; }, &sink);
    CHECK(st.ok() && sink.chunks.size() == 1 + w.rows.size(), "save: %zu chunks for %zu rows", sink.chunks.size(), w.rows.size());
    vk_index_params p2 = p;
    p2.n_shards = 2;
    p2.shard_devices[0] = devices.back();
    p2.shard_devices[1] = devices[0];
    struct Src { Sink *s; size_t at = 0; } src{&sink};
    std::unique_ptr<vk::Index> re;
    st = vk::load_sharded(p2, [](void *u, void *buf, uint64_t cap, uint64_t *len) -> int {
      Src *s = static_cast<Src *>(u);
      if (s->at >= s->s->chunks.size() || s->s->chunks[s->at].size() > cap) return 1;
      memcpy(buf, s->s->chunks[s->at].data(), s->s->chunks[s->at].size());
      *len = s->s->chunks[s->at++].size();
      return 0;
    }, &src, &re);
    CHECK(st.ok(), "load: %s", st.msg.c_str());
    if (st.ok()) {
      World w2;
      w2.devices = {devices.back(), devices[0]};
      w2.rows = w.rows;
      w2.ix = std::move(re);
      host_search(w2, "reloaded", 7, 10, nullptr, 0, nullptr, nullptr);
      w2.ix.reset();
    }
    report_violations(w, "save / load");
  }
  w.ix.reset();
  report_violations(w, "teardown");
}

// the pure layout arithmetic (shard_layout.hpp)
void check_layout(const std::vector<int> &devices) {
  const vk::ShardLayout l = vk::ShardLayout::from_devices(devices);
  const size_t S = devices.size(), nk = 37;
  std::vector<int> distinct;
  for (int d : devices)
    if (std::find(distinct.begin(), distinct.end(), d) == distinct.end()) distinct.push_back(d);
  CHECK(l.G() == distinct.size() && l.group_device == distinct, "groups in order of first appearance");
  CHECK(l.group_device[0] == devices[0] && l.groups[0][0] == 0, "group 0 is the serving device's");
  size_t maxp = 0;
  std::vector<char> recv(l.rccl_parts(), 0);
  for (size_t s = 0; s < S; ++s) {
    CHECK(l.group_device[l.group_of[s]] == devices[s], "shard %zu sits in its device's group", s);
    CHECK(l.groups[l.group_of[s]][l.pos_of[s]] == s, "position");
    CHECK(l.send_slot(s, nk) == l.pos_of[s] * nk && l.send_slot(s, nk) + nk <= l.send_entries(nk), "send slot inside the send buffer");
    CHECK(l.recv_slot(s, nk) % nk == 0 && l.recv_slot(s, nk) + nk <= l.recv_entries(nk), "receive slot inside the receive buffer");
    // what ncclAllGather does: rank r's send buffer lands at r * send_entries
    CHECK(l.recv_slot(s, nk) == l.group_of[s] * l.send_entries(nk) + l.send_slot(s, nk), "receive offset = rank * count + send offset");
    CHECK(!recv[l.recv_slot(s, nk) / nk], "two shards share a receive slot");
    recv[l.recv_slot(s, nk) / nk] = 1;
    CHECK(l.peer_slice(s, nk) == s * nk, "peer-copy slice");
    maxp = std::max<size_t>(maxp, l.pos_of[s] + 1);
  }
  CHECK(l.P == maxp, "P = the most shards any device holds");
  size_t pads = 0;
  for (size_t g = 0; g < l.G(); ++g) {
    for (size_t p = l.pad_begin(g); p < l.P; ++p) { CHECK(!recv[g * l.P + p], "a pad slot is nobody's"); ++pads; }
    CHECK(l.collective_lane(g) == l.groups[g].back(), "the collective goes on the last shard's lane");
  }
  CHECK(pads + S == l.rccl_parts(), "pads + shards = merge parts");
}

// the model must object when it should: its own known-bad cases
void model_self_test() {
  hipv::set_device_count(3);
  (void)hipSetDevice(0);
  hipStream_t s0, s1;
  (void)hipStreamCreateWithFlags(&s0, 0);
  void *a = nullptr, *b = nullptr;
  (void)hipMalloc(&a, 256);
  (void)hipSetDevice(1);
  (void)hipStreamCreateWithFlags(&s1, 0);
  (void)hipMalloc(&b, 256);
  (void)hipMemcpyPeerAsync(b, 1, a, 0, 256, s1);                    // peer access never enabled
  CHECK(hipv::take_violations().size() == 1, "model: peer copy without peer access");
  (void)hipMemcpyPeerAsync(b, 1, a, 0, 512, s1);                    // past the end (not executed by memmove: n is checked first)
  CHECK(hipv::take_violations().size() >= 1, "model: overrun");
  (void)hipStreamSynchronize(s1);
  hipv::launch(s1, "k", {{a, 16, "a"}}, [] {});                     // a kernel on device 1 touching device 0's memory
  CHECK(hipv::take_violations().size() == 1, "model: remote buffer in a kernel");
  hipEvent_t e;
  (void)hipSetDevice(0);
  (void)hipEventCreateWithFlags(&e, 0);
  CHECK(hipEventRecord(e, s1) != hipSuccess && hipv::take_violations().size() == 1, "model: event of device 0 on a stream of device 1");
  (void)hipEventDestroy(e);
  (void)hipStreamSynchronize(s0);
  (void)hipStreamSynchronize(s1);
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipStreamDestroy(s0);
  (void)hipStreamDestroy(s1);
}
}  // namespace

int main(int argc, char **argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 1;
  model_self_test();
  const std::vector<std::vector<int>> lists = {{0}, {0, 0, 0}, {0, 1}, {1, 0}, {0, 1, 2, 3}, {0, 1, 2, 3, 4, 5, 6, 7}, {0, 0, 1, 2, 2, 2}, {0, 1, 1},
                                               {3, 1, 3, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7}, {5, 5, 5, 2, 2, 7}};
  for (const auto &d : lists) check_layout(d);
  hipv::set_device_count(8);
  size_t worlds = 0;
  for (const auto &d : lists) {
    if (level < 2 && d.size() > 8) continue;
    run_world(d, /*hnsw=*/false, /*threads=*/true);
    ++worlds;
    if (level >= 2 || d.size() == 6 || d.size() == 4) { run_world(d, /*hnsw=*/true, /*threads=*/true); ++worlds; }
    if (d.size() == 3) { run_world(d, /*hnsw=*/false, /*threads=*/false); ++worlds; }   // shard-threads = 0: the caller enqueues every shard
  }
  // devices without a direct path: refused, loudly -- unless the deployment accepts staging through host memory
  {
    hipv::set_peer_capable(0, 1, false);
    vk_index_params p{};
    p.struct_size = sizeof p;
    p.algo = VK_ALGO_FLAT;
    p.dim = kDim;
    p.initial_cap = 100;
    p.n_shards = 2;
    p.shard_devices[0] = 0;
    p.shard_devices[1] = 1;
    std::unique_ptr<vk::Index> ix;
    unsetenv("VK_SHARD_ALLOW_STAGED");
    Status st = vk::create_sharded(p, &ix);
    CHECK(!st.ok() && st.code == VK_ERR_NO_DEVICE, "no peer access: %d %s", st.code, st.msg.c_str());
    setenv("VK_SHARD_ALLOW_STAGED", "1", 1);
    st = vk::create_sharded(p, &ix);
    CHECK(st.ok(), "shard-allow-staged: %s", st.msg.c_str());
    unsetenv("VK_SHARD_ALLOW_STAGED");
    ix.reset();
    hipv::set_peer_capable(0, 1, true);
    (void)hipv::take_violations();
  }
  printf("worlds=%zu kernels=%llu peer_copies=%llu collectives=%llu bad=%llu\n", worlds, (unsigned long long)hipv::kernels(),
         (unsigned long long)hipv::peer_copies(), (unsigned long long)hipv::collectives(), (unsigned long long)g_bad.load());
  return g_bad.load() ? 1 : 0;
}
