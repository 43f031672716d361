// CPU-side harness for csrc/coalescer.hpp (N1): a fake index whose batched search answers every query with
// a function of the query alone and records the batch sizes it was called with.  Many threads issue
// single-query requests through the Coalescer; each must get exactly its own answer, every request must be
// served exactly once, and concurrent requests must actually travel together.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "coalescer.hpp"

namespace {
struct FakeIndex final : vk::Index {
  explicit FakeIndex(const vk_index_params &p) : Index(p) {}
  std::atomic<uint64_t> calls{0}, queries{0}, max_batch{0};
  int delay_us = 200;   // a device pass takes a while: requests pile up behind it
  vk::Status search(const vk::SearchRequest &rq, float *od, uint64_t *ol, uint64_t *on) override {
    calls += 1;
    queries += rq.nq;
    uint64_t m = max_batch.load();
    while (rq.nq > m && !max_batch.compare_exchange_weak(m, rq.nq)) {}
    std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
    if (rq.k == 13) return vk::Status::Err(VK_ERR_INTERNAL, "k = 13 fails on purpose");
    for (uint64_t q = 0; q < rq.nq; ++q) {
      const float *v = rq.queries + q * params_.dim;
      const uint64_t n = rq.k < 3 ? rq.k : 3;            // "fewer than k found" is part of the contract
      on[q] = n;
      // a per-query filter shows in the answer: word 0 of the query's own bitmap is added to every label
      const uint64_t tag = rq.allow_tab && rq.allow_tab[q] ? rq.allow_tab[q][0] + rq.allow_nbits_tab[q] : 0;
      for (uint64_t i = 0; i < n; ++i) {
        od[q * rq.k + i] = v[0] * 1000.f + (float)i + (float)rq.ef * 0.001f;
        ol[q * rq.k + i] = (uint64_t)v[1] * 10 + i + tag;
      }
    }
    return vk::Status::Ok();
  }
  // not used by the coalescer
  vk::Status add(uint64_t, const float *) override { return vk::Status::Ok(); }
  vk::Status add_batch(const uint64_t *, const float *, uint64_t) override { return vk::Status::Ok(); }
  vk::Status remove(uint64_t) override { return vk::Status::Ok(); }
  vk::Status resize(uint64_t) override { return vk::Status::Ok(); }
  vk::Status set_ef(uint32_t) override { return vk::Status::Ok(); }
  vk::Status flush() override { return vk::Status::Ok(); }
  vk::Status search_device(const vk::SearchRequest &, float *, uint64_t *, uint32_t *, hipStream_t) override { return vk::Status::Ok(); }
  vk::Status label_distances(const float *, const uint64_t *, uint64_t, float *, uint8_t *) override { return vk::Status::Ok(); }
  vk::Status distance(uint64_t, const float *, float *) override { return vk::Status::Ok(); }
  vk::Status get_row(uint64_t, float *) override { return vk::Status::Ok(); }
  vk::Status contains(uint64_t, bool *) override { return vk::Status::Ok(); }
  vk::Status stats(vk_index_stats *) override { return vk::Status::Ok(); }
  vk::Status device_rows(uint64_t, void **, uint64_t *) override { return vk::Status::Ok(); }
  vk::Status commit_device_rows(uint64_t, const uint64_t *) override { return vk::Status::Ok(); }
  vk::Status save(vk_write_chunk_fn, void *) override { return vk::Status::Ok(); }
};
}  // namespace

// threads x per_thread requests, lanes (k, ef) taken round robin from `lanes` pairs; returns 0 when every
// request got its own answer.  out[0..3] = device calls, queries served, largest batch, coalescer batches.
extern "C" int coalescer_run(int threads, int per_thread, uint32_t max_batch, uint32_t max_wait_us, int n_lanes,
                             int with_failing_lane, uint64_t *out) {
  const bool toggle = (with_failing_lane & 2) != 0;
  with_failing_lane &= 1;
  vk_index_params p{};
  p.struct_size = sizeof p;
  p.dim = 4;
  FakeIndex ix(p);
  vk::Coalescer co;
  co.configure(max_batch, max_wait_us);
  std::atomic<int> bad{0};
  auto worker = [&](int t) {
    for (int r = 0; r < per_thread; ++r) {
      const int id = t * per_thread + r;
      const int lane = id % n_lanes;
      const bool fail = with_failing_lane && lane == n_lanes - 1;
      const uint64_t k = fail ? 13 : (uint64_t)(lane + 1);      // k = 1, 2, 3, ... (answers carry min(k,3) entries)
      const uint64_t ef = 100 + (uint64_t)lane;
      float q[4] = {(float)id, (float)(id + 7), 0.f, 0.f};
      float d[16];
      uint64_t l[16], n = 99;
      // every third request carries its own filter; every 17th is cancelled before it is served and must come back
      // at once with nothing, whatever the batch it would have ridden in does
      const bool filtered = id % 3 == 0, cancelled = !fail && id % 17 == 5;
      const uint64_t bits[1] = {(uint64_t)id * 1000};
      const uint64_t nbits = 64 + (uint64_t)(id % 5);
      volatile int flag = cancelled ? 1 : 0;
      if (toggle && id == threads * per_thread / 2) co.configure(0, 0);      // coalescing switched off under load
      vk::Status st = co.search(&ix, q, k, ef, filtered ? bits : nullptr, filtered ? nbits : 0, &flag, true, d, l, &n);
      if (cancelled) {
        if (!st.ok() || n != 0) bad += 1;
        continue;
      }
      if (fail) {
        if (st.ok() || st.msg.find("on purpose") == std::string::npos) bad += 1;
        continue;
      }
      const uint64_t want_n = k < 3 ? k : 3;
      if (!st.ok() || n != want_n) { bad += 1; continue; }
      for (uint64_t i = 0; i < n; ++i)
        if (d[i] != (float)id * 1000.f + (float)i + (float)ef * 0.001f ||
            l[i] != (uint64_t)(id + 7) * 10 + i + (filtered ? bits[0] + nbits : 0)) bad += 1;
    }
  };
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t) ts.emplace_back(worker, t);
  for (auto &t : ts) t.join();
  out[0] = ix.calls;
  out[1] = ix.queries;
  out[2] = ix.max_batch;
  out[3] = co.batches();
  out[4] = co.queries();
  return bad.load();
}
