// flat_index.cc -- FlatIndex: hnswlib::BruteforceSearch<float> semantics
// (third_party/hnswlib/bruteforce.h) over an HBM-resident row table.
//
//   addPoint     bruteforce.h:66-83    label known -> overwrite in place, else append,
//                                      "exceeds the specified limit" when full
//   removePoint  bruteforce.h:92-113   last element moves into the hole
//   searchKnn    bruteforce.h:116-145  -> flat_scan.hip (K3) + merge; result = the k
//                                      smallest by (distance,label)
//   resizeIndex  bruteforce.h:209-211
// Difference kept on purpose: with a filter, bruteforce.h:120-141 can return fewer than k
// allowed rows when fewer than k of the FIRST k rows pass (lastdist is then the max of
// those that passed).  That corner is unreachable through FT.SEARCH (FLAT + filter always
// takes the pre-filter path, src/query/planner.cc:23-29); here a filtered scan returns
// the exact k best allowed rows.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <queue>
#include <thread>

#include "index.hpp"

namespace vk {

// ---- contexts (DevBuf / PinBuf: index_common.cc) ------------------------------------------
Status SearchCtx::begin_on(hipStream_t s) {
  if (has_busy) {
    VK_HIP_TRY(hipStreamWaitEvent(s, busy, 0));
    has_busy = false;
  }
  return Status::Ok();
}

Status SearchCtx::end_async(hipStream_t s) {
  if (!busy) VK_HIP_TRY(hipEventCreateWithFlags(&busy, hipEventDisableTiming));
  VK_HIP_TRY(hipEventRecord(busy, s));
  has_busy = true;
  return Status::Ok();
}

Status SearchCtx::arm_cancel(const volatile int *caller_flag, const uint32_t **device_word, uint64_t n_members,
                             const uint32_t **member_words) {
  *device_word = nullptr;
  if (member_words) *member_words = nullptr;
  if (!caller_flag && (!n_members || !member_words)) return Status::Ok();
  if (!member_words) n_members = 0;
  VK_TRY(h_cancel.ensure((kMemberCancelOffset + n_members) * 4 + 64));
  volatile uint32_t *w = h_cancel.as<volatile uint32_t>();
  w[0] = 0u;   // (the waiting thread raises it; a flag that was up at entry is the caller's case)
  for (uint64_t i = 0; i < n_members; ++i) w[kMemberCancelOffset + i] = 0u;
  if (caller_flag) *device_word = h_cancel.as<uint32_t>();
  if (n_members) *member_words = h_cancel.as<uint32_t>() + kMemberCancelOffset;
  return Status::Ok();
}

Status SearchCtx::wait(const volatile int *caller_flag) { return wait(caller_flag, nullptr, 0); }

Status SearchCtx::wait(const volatile int *caller_flag, const volatile uint32_t *member_cancel, uint64_t nq) {
  if ((!caller_flag && !member_cancel) || !h_cancel.p) {
    VK_HIP_TRY(hipStreamSynchronize(stream));
    return Status::Ok();
  }
  volatile uint32_t *word = h_cancel.as<volatile uint32_t>();
  auto last_members = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(stream);
    if (e == hipSuccess) return Status::Ok();
    if (e != hipErrorNotReady) return Status::Err(4, std::string("hipStreamQuery: ") + hipGetErrorString(e));
    if (cancel_raised(caller_flag)) *word = 1u;
    if (member_cancel) {   // (a sweep over the members' words every 50 us: the batch may hold thousands)
      const auto now = std::chrono::steady_clock::now();
      if (now - last_members >= std::chrono::microseconds(50)) {
        last_members = now;
        for (uint64_t i = 0; i < nq; ++i)
          if (member_cancel[i] && !word[kMemberCancelOffset + i]) word[kMemberCancelOffset + i] = 1u;
      }
    }
    // (a short spin for the sub-millisecond searches, then naps: a runner that carries tokens -- every batch that comes through
    //  the adaptor does -- must not burn a core of a 16-CPU quota for the 5-15 ms of a full batch)
    if (spins < 300) __builtin_ia32_pause();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

SearchCtx::~SearchCtx() {
  for (TimedPair &t : timed) {
    if (t.t0) (void)hipEventDestroy(t.t0);
    if (t.t1) (void)hipEventDestroy(t.t1);
  }
  if (has_busy) (void)hipEventSynchronize(busy);
  if (busy) (void)hipEventDestroy(busy);
  if (stream) (void)hipStreamSynchronize(stream);
  for (DevBuf *b : {&d_q, &d_part_d, &d_part_l, &d_out_d, &d_out_l, &d_out_n, &d_allow, &d_idx, &d_tmp, &d_stats, &d_sync, &d_pool, &d_pool2, &d_redo,
                    &d_fq16, &d_fthr, &d_fcnt, &d_fcand, &d_fspill, &d_fsmax, &d_fpart_d, &d_fpart_l, &d_allow_tab})
    b->release();
  for (PinBuf *b : {&h_q, &h_out_d, &h_out_l, &h_out_n, &h_tmp, &h_idx, &h_cancel}) b->release();
  if (stream) (void)hipStreamDestroy(stream);
}

CtxPool::~CtxPool() {
  (void)hipSetDevice(device_);
  all_.clear();
}

SearchCtx *CtxPool::acquire(hipStream_t on) {
  SearchCtx *c = nullptr;
  {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (!free_.empty()) {
        c = free_.back();
        free_.pop_back();
        break;
      }
      if (all_.size() < max_) {
        (void)hipSetDevice(device_);
        auto n = std::make_unique<SearchCtx>();
        (void)hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
        all_.push_back(std::move(n));
        return all_.back().get();
      }
      cv_.wait(lk);
    }
  }
  if (c->has_busy) {
    (void)hipSetDevice(device_);
    if (!c->begin_on(on ? on : c->stream).ok()) {   // (could not order the streams: wait on the host instead)
      (void)hipEventSynchronize(c->busy);
      c->has_busy = false;
    }
  }
  return c;
}

void CtxPool::release(SearchCtx *c) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    free_.push_back(c);
  }
  cv_.notify_one();
}

Status upload_queries(SearchCtx *ctx, const float *queries, uint64_t nq, uint32_t dim, uint32_t stride_f, bool parallel,
                      const float *const *query_tab) {
  size_t bytes = (size_t)nq * stride_f * 4;
  VK_TRY(ctx->h_q.ensure(bytes));
  VK_TRY(ctx->d_q.ensure(bytes));
  float *h = ctx->h_q.as<float>();
  if (query_tab) {
    // one pointer per query (the dispatcher's batches: every query still lies in its caller's buffer) -- gathered straight
    // into the pinned staging block, no intermediate copy
    for (uint64_t q = 0; q < nq; ++q) {
      memcpy(h + q * stride_f, query_tab[q], (size_t)dim * 4);
      if (stride_f != dim) memset(h + q * stride_f + dim, 0, (size_t)(stride_f - dim) * 4);
    }
    VK_HIP_TRY(hipMemcpyAsync(ctx->d_q.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return Status::Ok();
  }
  if (parallel && stride_f == dim && bytes >= ((size_t)8 << 20)) {
    // a large batch (8192 x 768 queries = 25 MB): the copy into pinned memory, not the DMA, is what takes the time (3 ms on
    // one thread) -- four threads copy a quarter each and hand their pieces to the copy engine as they finish them
    const size_t piece = (size_t)2 << 20;
    const int nt = 4;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t per = ((bytes / nt) + piece - 1) / piece * piece;
    std::vector<std::thread> ts;
    std::mutex mu;
    hipError_t first_err = hipSuccess;
    for (int t = 0; t < nt; ++t)
      ts.emplace_back([&, t] {
        (void)hipSetDevice(dev);
        const size_t lo = std::min(bytes, (size_t)t * per), hi = std::min(bytes, lo + per);
        for (size_t o = lo; o < hi; o += piece) {
          const size_t n = std::min(piece, hi - o);
          memcpy(reinterpret_cast<char *>(h) + o, reinterpret_cast<const char *>(queries) + o, n);
          std::lock_guard<std::mutex> g(mu);
          const hipError_t e = hipMemcpyAsync(static_cast<char *>(ctx->d_q.p) + o, reinterpret_cast<char *>(h) + o, n, hipMemcpyHostToDevice, ctx->stream);
          if (e != hipSuccess && first_err == hipSuccess) first_err = e;
        }
      });
    for (auto &t : ts) t.join();
    VK_HIP_TRY(first_err);
    return Status::Ok();
  }
  if (stride_f == dim) {
    memcpy(h, queries, bytes);
  } else {
    for (uint64_t q = 0; q < nq; ++q) {
      memcpy(h + q * stride_f, queries + q * dim, (size_t)dim * 4);
      memset(h + q * stride_f + dim, 0, (size_t)(stride_f - dim) * 4);
    }
  }
  VK_HIP_TRY(hipMemcpyAsync(ctx->d_q.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
  return Status::Ok();
}

Status upload_allow(SearchCtx *ctx, const uint64_t *allow_bits, uint64_t allow_nbits, const uint64_t **d_allow) {
  *d_allow = nullptr;
  if (!allow_bits) return Status::Ok();
  size_t words = (size_t)((allow_nbits + 63) / 64);
  VK_TRY(ctx->d_allow.ensure(std::max<size_t>(words * 8, 8)));
  if (words) VK_HIP_TRY(hipMemcpyAsync(ctx->d_allow.p, allow_bits, words * 8, hipMemcpyHostToDevice, ctx->stream));
  *d_allow = ctx->d_allow.as<uint64_t>();
  return Status::Ok();
}

// vector_base.cc:509-530: fill to k, then a key replaces the heap top only when its
// distance is strictly smaller (ties keep what is already there).
void prefilter_heap_select(const float *dist, const uint64_t *labels, uint64_t n, uint64_t k, float *out_dist,
                           uint64_t *out_label, uint64_t *out_n) {
  std::priority_queue<std::pair<float, uint64_t>> results;
  for (uint64_t i = 0; i < n; ++i) {
    if (labels[i] == ~0ull) continue;  // unknown key: ComputeDistanceFromRecord failed
    if (results.size() < k) {
      results.emplace(dist[i], labels[i]);
    } else if (k && dist[i] < results.top().first) {
      results.pop();
      results.emplace(dist[i], labels[i]);
    }
  }
  uint64_t m = results.size();
  *out_n = m;
  while (m) {
    --m;
    out_dist[m] = results.top().first;
    out_label[m] = results.top().second;
    results.pop();
  }
}

// the pre-filter path on any index: per-key distances from the device, then the reference's heap rule in key order
Status Index::search_labels(const float *query, uint64_t k, const uint64_t *labels, uint64_t n, float *out_dist,
                            uint64_t *out_label, uint64_t *out_n) {
  *out_n = 0;
  if (n == 0 || k == 0) return Status::Ok();
  std::vector<float> d(n);
  std::vector<uint8_t> found(n);
  VK_TRY(label_distances(query, labels, n, d.data(), found.data()));
  std::vector<uint64_t> lab(n);
  for (uint64_t i = 0; i < n; ++i) lab[i] = found[i] ? labels[i] : ~0ull;   // unknown key: skipped by the heap
  prefilter_heap_select(d.data(), lab.data(), n, k, out_dist, out_label, out_n);
  return Status::Ok();
}

#ifdef VK_EXPERIMENTS
// margin audit of the candidate filter (experiments build only; tests/helpers/exp_margin_check.py): device buffers the
// final pass of every FLAT index of the process dumps its gate's view into.  Process-wide on purpose -- one test, one index.
struct ExpFilterDump { float *scores = nullptr, *thr = nullptr, *qstate = nullptr; uint32_t rows = 0, ld = 0; };   // qstate: [6][ld] = c2, c1, c0, closed | L_q | margin at the norm cap
static ExpFilterDump g_exp_dump;
static unsigned long long *g_exp_rerank_stamps = nullptr;   // [nq][16], device memory (vk_exp_rerank_stamps)
}  // namespace vk
extern "C" __attribute__((visibility("default"))) void vk_exp_rerank_stamps(unsigned long long *d_stamps) { vk::g_exp_rerank_stamps = d_stamps; }
extern "C" __attribute__((visibility("default"))) void vk_exp_filter_dump(float *d_scores, float *d_thr, float *d_qstate, uint32_t rows, uint32_t ld) {
  vk::g_exp_dump.scores = d_scores;
  vk::g_exp_dump.thr = d_thr;
  vk::g_exp_dump.qstate = d_qstate;
  vk::g_exp_dump.rows = rows;
  vk::g_exp_dump.ld = ld;
}
namespace vk {
static bool filter_dump_active() { return g_exp_dump.scores != nullptr; }   // (the audit holds ONE bound per query against the gate: one pass)
#else
static constexpr bool filter_dump_active() { return false; }
#endif

// ---- FlatIndex ---------------------------------------------------------------------------
class FlatIndex final : public Index {
 public:
  explicit FlatIndex(const vk_index_params &p, int device)
      : Index(p), store_(device, p.dim, p.dtype == VK_DTYPE_BF16), pool_(device), capacity_(p.initial_cap) {}
  ~FlatIndex() override {
    (void)hipSetDevice(store_.device());
    d_rowstats_.release();
    d_hn16_.release();
    d_tile_norm_.release();
  }

  Status add(uint64_t label, const float *row) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    return add_locked(label, row);
  }

  Status add_batch(const uint64_t *labels, const float *rows, uint64_t n) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    // Runs of labels that are new take consecutive slots at the end of the table (bruteforce.h:66-83), so a run
    // can go from the caller's rows to HBM in one strided copy instead of row by row through the staging log;
    // everything else (updates, the element that hits the capacity limit) keeps addPoint's path and message.
    std::vector<uint64_t> run_labels;
    uint64_t run_begin = 0;
    auto close_run = [&](uint64_t end) -> Status {
      if (run_labels.empty()) return Status::Ok();
      const uint32_t first = (uint32_t)(count_ - run_labels.size());
      Status s = store_.bulk_write(first, rows + run_begin * params_.dim, end - run_begin, run_labels.data());
      if (!s.ok()) {   // (the table could not grow: the run never reached it -- its labels and slots are given back, so the
                       //  index keeps answering over what it holds and the caller may send the batch again)
        for (uint64_t l : run_labels) slot_of_.erase(l);
        count_ -= run_labels.size();
      }
      run_labels.clear();
      return s;
    };
    const bool bulk = n >= 256 && !store_.bf16();
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t label = labels ? labels[i] : i;
      if (bulk && count_ < capacity_ && slot_of_.find(label) == slot_of_.end()) {
        if (run_labels.empty()) run_begin = i;
        slot_of_.emplace(label, (uint32_t)count_++);
        run_labels.push_back(label);
        if (label > max_label_) max_label_ = label;   // ("ever held": a run that is given back below leaves it, like a removed label does)
        continue;
      }
      VK_TRY(close_run(i));
      VK_TRY(add_locked(label, rows + i * params_.dim));
      if (store_.staged_bytes() >= ((size_t)256 << 20)) VK_TRY(store_.flush());
    }
    return close_run(n);
  }

  Status remove(uint64_t label) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    auto it = slot_of_.find(label);
    if (it == slot_of_.end()) return Status::Ok();  // bruteforce.h:95-98: silently ignored
    uint32_t cur = it->second;
    slot_of_.erase(it);
    uint32_t last = (uint32_t)count_ - 1;
    if (cur != last) {
      uint64_t moved = store_.host_labels()[last];
      slot_of_[moved] = cur;
      store_.stage_move(cur, last, moved);
    }
    count_--;
    return Status::Ok();
  }

  Status resize(uint64_t new_max) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    capacity_ = new_max;
    return Status::Ok();
  }

  Status set_ef(uint32_t) override { return Status::Ok(); }

  // (nothing staged: no exclusive lock, so a flush at every write -> read switch does not wait for the batches in flight)
  Status flush() override { return flush_if_dirty(); }

  Status search(const SearchRequest &rq_in, float *out_dist, uint64_t *out_label, uint64_t *out_n) override {
    SearchRequest rq = rq_in;
    const uint64_t *d_filter = nullptr;
    if (rq.filter) {   // a device-resident filter (filter_set.hpp): its bitmap is used where it lies
      d_filter = rq.filter->bits_on(store_.device());
      if (!d_filter) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's device");
      rq.allow_bits = nullptr;
      rq.allow_nbits = rq.filter->nbits();
    }
    // (FLAT + filter is the pre-filter path in valkey-search, planner.cc:23-29: per-query bitmaps are served run by run)
    if (rq.allow_tab || rq.filter_tab) return search_grouped_by_filter(this, rq, out_dist, out_label, out_n);
    VK_TRY(flush_if_dirty());
    std::shared_lock<std::shared_mutex> lk(rw_);
    (void)hipSetDevice(store_.device());
    const uint64_t count = count_;
    const uint64_t k = std::min<uint64_t>(rq.k, count);  // vector_flat.cc:234-236
    if (k == 0 || rq.nq == 0) {
      for (uint64_t q = 0; q < rq.nq; ++q) out_n[q] = 0;
      return Status::Ok();
    }
    CtxLease lease(pool_);
    SearchCtx *ctx = lease.ctx;
    filter_used_ = false;
    VK_TRY(upload_queries(ctx, rq.queries, rq.nq, params_.dim, store_.stride_f(), opt_.get(kOptUploadParallel) != 0, rq.query_tab));
    const uint64_t *d_allow = d_filter;
    if (!d_filter) VK_TRY(upload_allow(ctx, rq.allow_bits, rq.allow_nbits, &d_allow));
    if (k > kMaxPassK) return search_in_passes(ctx, rq, k, count, d_allow, out_dist, out_label, out_n);
    VK_TRY(ctx->h_out_d.ensure(rq.nq * k * 4));
    VK_TRY(ctx->h_out_l.ensure(rq.nq * k * 8));
    VK_TRY(ctx->h_out_n.ensure(rq.nq * 4));
    if (rq.nq * k <= kZeroCopyEntries) {
      VK_TRY(scan(ctx, ctx->d_q.as<float>(), rq.nq, k, count, d_allow, rq.allow_nbits, rq.cancel_flag,
                  ctx->h_out_d.as<float>(), ctx->h_out_l.as<uint64_t>(), ctx->h_out_n.as<uint32_t>(), ctx->stream));
    } else {
      VK_TRY(ctx->d_out_d.ensure(rq.nq * k * 4));
      VK_TRY(ctx->d_out_l.ensure(rq.nq * k * 8));
      VK_TRY(ctx->d_out_n.ensure(rq.nq * 4));
      VK_TRY(scan(ctx, ctx->d_q.as<float>(), rq.nq, k, count, d_allow, rq.allow_nbits, rq.cancel_flag,
                  ctx->d_out_d.as<float>(), ctx->d_out_l.as<uint64_t>(), ctx->d_out_n.as<uint32_t>(), ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_d.p, ctx->d_out_d.p, rq.nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_l.p, ctx->d_out_l.p, rq.nq * k * 8, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_n.p, ctx->d_out_n.p, rq.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (filter_used_) {   // survivor counts + the number of handed-over queries of the candidate filter, for vk_index_stats
      // (layout of the batch's words: scan_filter -- [nq] counts | [nq] flags | spill_next, redo_cnt, ...)
      // ... | [nq] redo list | [nq][kSpillPerQuery] chunk slots | [nq] arrival counters | [nq] rows the re-rank evaluated
      VK_TRY(ctx->h_tmp.ensure(rq.nq * 12 + 8));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_tmp.p, ctx->d_fcnt.p, rq.nq * 8 + 8, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_tmp.as<char>() + rq.nq * 8 + 8, ctx->d_fcnt.as<uint32_t>() + (4 + 4 * rq.nq + rq.nq * kSpillPerQuery),
                                rq.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    VK_TRY(ctx->wait(rq.cancel_flag));   // (a raised flag stops the kernels: the answer is what they had, bruteforce.h:129)
    if (filter_used_) {
      const uint32_t *c = ctx->h_tmp.as<uint32_t>();
      uint64_t sum = 0;
      for (uint64_t q = 0; q < rq.nq; ++q) sum += c[q];
      last_filter_cands_ = sum;
      last_filter_fallback_ = c[2 * rq.nq + 1];   // queries the exact redo pass answered
      uint64_t rr = 0;
      for (uint64_t q = 0; q < rq.nq; ++q) rr += c[2 * rq.nq + 2 + q];
      last_filter_reranked_ = rr;                 // rows that got an exact distance
    } else {
      last_filter_cands_ = 0;
      last_filter_fallback_ = 0;
      last_filter_reranked_ = 0;
    }
    // caller's buffers are [nq][rq.k]
    for (uint64_t q = 0; q < rq.nq; ++q) {
      uint32_t n = ctx->h_out_n.as<uint32_t>()[q];
      out_n[q] = n;
      memcpy(out_dist + q * rq.k, ctx->h_out_d.as<float>() + q * k, (size_t)n * 4);
      memcpy(out_label + q * rq.k, ctx->h_out_l.as<uint64_t>() + q * k, (size_t)n * 8);
    }
    return Status::Ok();
  }

  Status search_device(const SearchRequest &rq, float *d_out_dist, uint64_t *d_out_label, uint32_t *d_out_n,
                       hipStream_t stream) override {
    VK_TRY(flush_if_dirty());
    std::shared_lock<std::shared_mutex> lk(rw_);
    (void)hipSetDevice(store_.device());
    if (rq.nq == 0) return Status::Ok();
    if (rq.k == 0) return Status::Err(VK_ERR_INVALID, "search_batch_device needs k > 0");
    // a context per call (concurrent callers never share scratch); the work goes on the caller's stream and is still
    // in flight when the call returns -- the context's next user is ordered behind it (SearchCtx::busy)
    CtxLease lease(pool_, stream);
    SearchCtx *ctx = lease.ctx;
    hipStream_t s = stream ? stream : ctx->stream;
    const uint64_t count = count_;
    // a shard with fewer than k rows (vector_flat.cc:234-236 clamps k): the lists are [nq][rq.k] all the same, the
    // tail is (+inf, UINT64_MAX) -- what the shard merge expects of a short shard
    const uint64_t k = std::min<uint64_t>(rq.k, count);
    if (k == 0) {
      VK_HIP_TRY(launch_fill_empty(d_out_dist, d_out_label, d_out_n, (uint32_t)rq.nq, (uint32_t)rq.k, s));
      return ctx->end_async(s);
    }
    const float *dq = rq.queries;
    if (store_.stride_f() != params_.dim) {
      const size_t q_pitch = (size_t)store_.stride_f() * 4;   // queries are always f32
      VK_TRY(ctx->d_q.ensure(rq.nq * q_pitch));
      VK_HIP_TRY(hipMemsetAsync(ctx->d_q.p, 0, rq.nq * q_pitch, s));
      VK_HIP_TRY(hipMemcpy2DAsync(ctx->d_q.p, q_pitch, rq.queries, (size_t)params_.dim * 4,
                                  (size_t)params_.dim * 4, rq.nq, hipMemcpyDeviceToDevice, s));
      dq = ctx->d_q.as<float>();
    }
    Status st = scan(ctx, dq, rq.nq, k, count, rq.allow_bits, rq.allow_nbits, nullptr, d_out_dist,
                     d_out_label, d_out_n, s, rq.k, rq.cancel_word);
    Status en = ctx->end_async(s);
    return st.ok() ? en : st;
  }

  Status label_distances(const float *query, const uint64_t *labels, uint64_t n, float *out_dist, uint8_t *found) override {
    VK_TRY(flush_if_dirty());
    std::shared_lock<std::shared_mutex> lk(rw_);
    (void)hipSetDevice(store_.device());
    if (n == 0) return Status::Ok();
    CtxLease lease(pool_);
    SearchCtx *ctx = lease.ctx;
    // label -> slot on the host (dict_external_to_internal, vector_flat.cc:260-265)
    VK_TRY(ctx->h_idx.ensure(n * 4));
    VK_TRY(ctx->h_tmp.ensure(n * 4));
    uint32_t *idx = ctx->h_idx.as<uint32_t>();
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) {
      auto it = slot_of_.find(labels[i]);
      found[i] = it != slot_of_.end();
      if (found[i]) idx[m++] = it->second;
    }
    if (m == 0) return Status::Ok();
    VK_TRY(upload_queries(ctx, query, 1, params_.dim, store_.stride_f()));
    VK_TRY(ctx->d_idx.ensure(m * 4));
    VK_TRY(ctx->d_tmp.ensure(m * 4));
    VK_HIP_TRY(hipMemcpyAsync(ctx->d_idx.p, idx, m * 4, hipMemcpyHostToDevice, ctx->stream));
    GatherArgs ga{store_.d_rows(), ctx->d_q.as<float>(), ctx->d_idx.as<uint32_t>(), ctx->d_tmp.as<float>(),
                  store_.stride_f(), store_.stride_f() / 16, (uint32_t)m};
    VK_HIP_TRY(launch_gather_distance(ga, l2(), store_.bf16(), ctx->stream));
    VK_HIP_TRY(hipMemcpyAsync(ctx->h_tmp.p, ctx->d_tmp.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const float *hd = ctx->h_tmp.as<float>();
    for (uint64_t i = 0, j = 0; i < n; ++i)
      if (found[i]) out_dist[i] = hd[j++];
    return Status::Ok();
  }

  Status distance(uint64_t label, const float *query, float *out) override {
    float d;
    uint64_t l, n = 0;
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      if (!slot_of_.count(label)) return Status::Err(VK_ERR_NOT_FOUND, "Couldn't find internal id");
    }
    VK_TRY(search_labels(query, 1, &label, 1, &d, &l, &n));
    if (n != 1) return Status::Err(VK_ERR_NOT_FOUND, "Couldn't find internal id");
    *out = d;
    return Status::Ok();
  }

  Status get_row(uint64_t label, float *out) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    auto it = slot_of_.find(label);
    if (it == slot_of_.end()) return Status::Err(VK_ERR_NOT_FOUND, "label not found");
    return store_.read_row(it->second, out);
  }

  Status contains(uint64_t label, bool *found) override {
    std::shared_lock<std::shared_mutex> lk(rw_);
    *found = slot_of_.count(label) != 0;
    return Status::Ok();
  }

  Status stats(vk_index_stats *out) override {
    std::shared_lock<std::shared_mutex> lk(rw_);
    memset(out, 0, sizeof(*out));
    out->count = count_;
    out->capacity = capacity_;
    out->device_bytes = store_.device_bytes();
    out->host_bytes = store_.host_bytes() + slot_of_.size() * 24;
    out->staged_ops = store_.staged_ops();
    out->max_level = -1;
    out->last_filter_candidates = last_filter_cands_;
    out->last_filter_fallback = last_filter_fallback_;
    out->last_filter_reranked = last_filter_reranked_;
    out->last_filter_final_rows = last_filter_final_rows_.load(std::memory_order_relaxed);
    out->max_label = max_label_;
    (void)hipSetDevice(store_.device());
    pool_.for_each_free([&](SearchCtx *c) { for (auto &tp : c->timed) drain_timed(tp); });
    // (with kernel-timing on the time covers exactly the batches counted; off: batches are still counted, the time stands still)
    out->filter_batches = filter_batches_;
    out->filter_kernel_ns = filter_ns_total_;
    return Status::Ok();
  }

  void filter_devices(std::vector<int> *out) const override { out->assign(1, store_.device()); }

  Status device_rows(uint64_t n, void **d_rows, uint64_t *stride_bytes) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    if (count_ != 0) return Status::Err(VK_ERR_INVALID, "device bulk load needs an empty index");
    if (n > capacity_) return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
    VK_TRY(store_.flush());
    VK_TRY(store_.reserve(n));
    *d_rows = store_.d_rows_mut();
    *stride_bytes = store_.row_bytes();
    return Status::Ok();
  }

  Status commit_device_rows(uint64_t n, const uint64_t *labels) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    if (count_ != 0) return Status::Err(VK_ERR_INVALID, "device bulk load needs an empty index");
    if (n > store_.alloc_rows()) return Status::Err(VK_ERR_INVALID, "commit exceeds the reserved rows");
    slot_of_.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
      uint64_t lab = labels ? labels[i] : i;
      if (lab > max_label_) max_label_ = lab;
      if (!slot_of_.emplace(lab, (uint32_t)i).second) {
        slot_of_.clear();
        return Status::Err(VK_ERR_INVALID, "duplicate label in bulk load");
      }
      store_.stage_label((uint32_t)i, lab);
    }
    count_ = n;
    store_.note_written(0, n);   // (the caller filled the rows on the device)
    return store_.flush();
  }

  Status save(vk_write_chunk_fn fn, void *user) override;
  Status load_from(vk_read_chunk_fn fn, void *user);

 private:
  Status add_locked(uint64_t label, const float *row) {
    auto it = slot_of_.find(label);
    uint32_t slot;
    if (it != slot_of_.end()) {
      slot = it->second;
    } else {
      if (count_ >= capacity_)
        return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
      slot = (uint32_t)count_++;
      slot_of_.emplace(label, slot);
      if (label > max_label_) max_label_ = label;
    }
    return store_.stage_write(slot, row, label);
  }

  Status flush_if_dirty() {
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      if (!store_.dirty()) return Status::Ok();
    }
    std::unique_lock<std::shared_mutex> lk(rw_);
    return store_.flush();
  }

  // enqueue scan + merge of rows [0,count) on stream s; all pointers device
  Status scan(SearchCtx *ctx, const float *d_q, uint64_t nq, uint64_t k, uint64_t count, const uint64_t *d_allow,
              uint64_t allow_nbits, const volatile int *cancel, float *d_out_d, uint64_t *d_out_l,
              uint32_t *d_out_n, hipStream_t s, uint64_t out_ld = 0, const uint32_t *cancel_word = nullptr) {
    if (out_ld == 0) out_ld = k;                      // entries per query in the output arrays (>= k; the tail is padding)
    int e = flat_scan_slots_per_lane(k);
    if (e == 0) return Status::Err(VK_ERR_INVALID, "k > 1024 needs the host entry points (vk_index_search / _batch), which page through the result in passes");
    // K4h + exact re-rank: a batch large enough that the exact matrix-core kernel is the bottleneck, an index large
    // enough that the pre-pass sample is a small part of it
    if (!lb_dist_ && nq >= filter_min_queries_ && !cancel_raised(cancel) && !force_scan_ && filter_enabled_ &&
        flat_filter_supported(store_.stride_f(), k, store_.bf16(), l2()) &&
        count >= 8 * filter_prepass_rows(k) && count >= filter_min_rows_) {
      VK_TRY(ensure_row_stats());
      // (an index that is mostly tiles the f16 pipe cannot carry -- un-normalised rows with values beyond 32768 -- would
      // send every pair to the re-rank: the exact kernels below serve it)
      if ((uint64_t)filter_bad_tiles_.load(std::memory_order_relaxed) * 64 <= count / 128) {
        const uint32_t *d_cancel = cancel_word;
        if (!d_cancel) VK_TRY(ctx->arm_cancel(cancel, &d_cancel));
        return scan_filter(ctx, d_q, nq, k, count, d_allow, allow_nbits, d_out_d, d_out_l, d_out_n, s, out_ld, d_cancel);
      }
    }
    // K4: enough queries to feed the matrix cores, inner-product space (IP / COSINE)
    if (!l2() && !lb_dist_ && nq >= kGemmMinQueries && !cancel_raised(cancel) && flat_gemm_supported(store_.stride_f(), k) && !force_scan_) {
      const uint32_t *d_cancel = cancel_word;
      if (!d_cancel) VK_TRY(ctx->arm_cancel(cancel, &d_cancel));
      return scan_gemm(ctx, d_q, nq, k, count, d_allow, allow_nbits, d_out_d, d_out_l, d_out_n, s, out_ld, d_cancel);
    }
    const uint32_t *d_cancel = cancel_word;
    uint64_t row_end = count;
    if (cancel_raised(cancel)) row_end = std::min<uint64_t>(count, k);          // (those rows are scanned whatever the flag says)
    else if (!d_cancel) VK_TRY(ctx->arm_cancel(cancel, &d_cancel));
    return scan_k3(ctx, d_q, nq, k, row_end, d_allow, allow_nbits, d_cancel, d_out_d, d_out_l, d_out_n, s, out_ld);
  }

  // K3 (+ merge) over rows [0, row_end): the VALU scan; run_flag: device-side conditional launch (kernels.hpp)
  Status scan_k3(SearchCtx *ctx, const float *d_q, uint64_t nq, uint64_t k, uint64_t row_end, const uint64_t *d_allow,
                 uint64_t allow_nbits, const uint32_t *d_cancel, float *d_out_d, uint64_t *d_out_l, uint32_t *d_out_n,
                 hipStream_t s, uint64_t out_ld, const uint32_t *run_flag = nullptr, uint32_t run_if = 0, uint32_t run_hi = 0,
                 const uint32_t *q_index = nullptr) {
    // q_index != nullptr: redo mode -- nq is the most queries the launch serves, *run_flag of them are listed in q_index
    int e = flat_scan_slots_per_lane(k);
    const uint32_t chunks = store_.stride_f() / 16;
    if ((size_t)chunks * 64 > 160 * 1024) return Status::Err(VK_ERR_INVALID, "dimension too large for the LDS query block");
    if (lb_dist_) e = 16;            // the paged scan has one instantiation: 16 slots per lane, one query per pass
    const int qb = lb_dist_ ? 1 : flat_scan_pick_qb(nq, chunks, e, l2());
    const uint32_t nqg = (uint32_t)((nq + qb - 1) / qb);
    // (rows: cancelled at entry -> only the first k rows are looked at, bruteforce.h:120-129; cancelled later -> the
    // kernel stops between row tiles, FlatScanArgs::cancel, and the answer is what the lists hold)
    // row partitions: enough blocks to fill 256 CUs, never more waves than 16-row tiles
    const uint64_t tiles = (row_end + 15) / 16;
    // ... and at least ~4 tiles per wave: every block leaves a partial list for the merge kernel (one wave per
    // query), which dominates the latency of a small index when there are thousands of near-empty lists
    uint32_t nrp = (uint32_t)std::min<uint64_t>((tiles + 15) / 16, std::max<uint32_t>((2048 + nqg - 1) / nqg, scan_min_nrp_));
    nrp = std::max<uint32_t>(8, (nrp + 7) & ~7u);
    const uint64_t per_q = (uint64_t)nrp * (e == 1 ? 1 : 4) * k;   // e == 1: one list per block (merged in LDS)
    VK_TRY(ctx->d_part_d.ensure((size_t)nq * per_q * 4));
    VK_TRY(ctx->d_part_l.ensure((size_t)nq * per_q * 8));
    const uint32_t done = 1;
    {
      FlatScanArgs a{};
      a.rows = store_.d_rows();
      a.labels = store_.d_labels();
      a.queries = d_q;
      a.allow_bits = d_allow;
      a.allow_nbits = allow_nbits;
      a.lb_dist = lb_dist_;
      a.lb_label = lb_label_;
      a.part_dist = ctx->d_part_d.as<float>();
      a.part_label = ctx->d_part_l.as<uint64_t>();
      a.row_stride_f = a.q_stride_f = store_.stride_f();
      a.chunks = chunks;
      a.row_begin = 0;
      a.row_end = (uint32_t)row_end;
      a.nq = (uint32_t)nq;
      a.k = (uint32_t)k;
      a.nrp = nrp;
      a.nqg = nqg;
      a.cancel = d_cancel;
      a.run_flag = run_flag;
      a.run_if = run_if;
      a.run_hi = run_hi;
      a.q_index = q_index;
      a.nq_dev = q_index ? run_flag : nullptr;
      VK_HIP_TRY(launch_flat_scan(a, l2(), store_.bf16(), qb, e, s));
    }
    MergeArgs m{};
    m.in_dist = ctx->d_part_d.as<float>();
    m.in_label = ctx->d_part_l.as<uint64_t>();
    m.part_stride = nq * per_q;
    m.q_stride = per_q;
    m.parts = done;
    m.per_part = (uint32_t)per_q;
    m.k = (uint32_t)k;
    m.out_ld = (uint32_t)out_ld;
    m.out_dist = d_out_d;
    m.out_label = d_out_l;
    m.out_n = d_out_n;
    m.run_flag = run_flag;
    m.run_if = run_if;
    m.run_hi = run_hi;
    m.q_index = q_index;
    m.nq_dev = q_index ? run_flag : nullptr;
    VK_HIP_TRY(launch_merge_topk(m, e, nq, s));
    return Status::Ok();
  }

  // k > 1024 (max-vector-knn allows up to 100000, ft_search_parser.cc:34-45): passes of <= 1024, each pass
  // restricted to entries strictly beyond the last (distance,label) of the previous one.  Exact; one
  // full scan per pass.
  Status search_in_passes(SearchCtx *ctx, const SearchRequest &rq, uint64_t k, uint64_t count, const uint64_t *d_allow,
                          float *out_dist, uint64_t *out_label, uint64_t *out_n) {
    const uint64_t nq = rq.nq;
    VK_TRY(ctx->d_out_d.ensure(nq * kMaxPassK * 4));
    VK_TRY(ctx->d_out_l.ensure(nq * kMaxPassK * 8));
    VK_TRY(ctx->d_out_n.ensure(nq * 4));
    VK_TRY(ctx->h_out_d.ensure(nq * kMaxPassK * 4));
    VK_TRY(ctx->h_out_l.ensure(nq * kMaxPassK * 8));
    VK_TRY(ctx->h_out_n.ensure(nq * 4));
    VK_TRY(ctx->d_idx.ensure(nq * 4));    // lower-bound distances
    VK_TRY(ctx->d_stats.ensure(nq * 8));  // lower-bound labels
    std::vector<float> lbd(nq, -__builtin_inff());
    std::vector<uint64_t> lbl(nq, 0);
    std::vector<uint64_t> got(nq, 0);
    bool first = true, more = true;
    while (more) {
      uint64_t kp = 0;
      for (uint64_t q = 0; q < nq; ++q) kp = std::max(kp, std::min<uint64_t>(kMaxPassK, k - got[q]));
      if (kp == 0) break;
      if (!first) {
        VK_HIP_TRY(hipMemcpyAsync(ctx->d_idx.p, lbd.data(), nq * 4, hipMemcpyHostToDevice, ctx->stream));
        VK_HIP_TRY(hipMemcpyAsync(ctx->d_stats.p, lbl.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream));
      }
      lb_dist_ = first ? nullptr : ctx->d_idx.as<float>();
      lb_label_ = first ? nullptr : ctx->d_stats.as<uint64_t>();
      Status st = scan(ctx, ctx->d_q.as<float>(), nq, kp, count, d_allow, rq.allow_nbits, nullptr, ctx->d_out_d.as<float>(),
                       ctx->d_out_l.as<uint64_t>(), ctx->d_out_n.as<uint32_t>(), ctx->stream);
      lb_dist_ = nullptr;
      lb_label_ = nullptr;
      VK_TRY(st);
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_d.p, ctx->d_out_d.p, nq * kp * 4, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_l.p, ctx->d_out_l.p, nq * kp * 8, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_n.p, ctx->d_out_n.p, nq * 4, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipStreamSynchronize(ctx->stream));
      more = false;
      for (uint64_t q = 0; q < nq; ++q) {
        const uint64_t want = std::min<uint64_t>(kp, k - got[q]);
        const uint64_t n = std::min<uint64_t>(ctx->h_out_n.as<uint32_t>()[q], want);
        memcpy(out_dist + q * rq.k + got[q], ctx->h_out_d.as<float>() + q * kp, n * 4);
        memcpy(out_label + q * rq.k + got[q], ctx->h_out_l.as<uint64_t>() + q * kp, n * 8);
        got[q] += n;
        if (n) { lbd[q] = out_dist[q * rq.k + got[q] - 1]; lbl[q] = out_label[q * rq.k + got[q] - 1]; }
        if (n == kp && got[q] < k) more = true;   // a full pass: there may be more beyond it
      }
      first = false;
    }
    for (uint64_t q = 0; q < nq; ++q) out_n[q] = got[q];
    return Status::Ok();
  }

  // K4 launch: persistent grid of ~one block per CU, nrp row partitions x nqt query tiles of 32
  Status scan_gemm(SearchCtx *ctx, const float *d_q, uint64_t nq, uint64_t k, uint64_t count, const uint64_t *d_allow,
                   uint64_t allow_nbits, float *d_out_d, uint64_t *d_out_l, uint32_t *d_out_n, hipStream_t s,
                   uint64_t out_ld = 0, const uint32_t *d_cancel = nullptr, const float *bound_given = nullptr,
                   const uint32_t *run_flag = nullptr, uint32_t run_if = 0, uint32_t run_hi = 0) {
    if (out_ld == 0) out_ld = k;
    // run_flag != nullptr: the candidate filter's whole-batch hand-over (k <= 10: lists in registers).  It is enqueued
    // behind every filtered batch and almost never runs, so it takes nothing that would cost a launch of its own: no
    // pre-pass bound, no lockstep window (whose progress words need clearing), no shared bound word.
    const bool handover = run_flag != nullptr;
    // pre-pass (this kernel over the first rows): a valid bound on every query's k-th best distance, so the per-lane
    // lists of K4 start gated instead of accepting everything until they have filled
    const float *init_bound = bound_given;   // (the candidate-filter path has run its own, larger, pre-pass)
    // (more rows for a larger k: the bound is the k-th best of the sample, and the lists of K4 pay per insert)
    const uint64_t pre_rows = gemm_prepass_rows_ * ((k + 9) / 10);
    if (!bound_given && pre_rows && count >= 8 * pre_rows && !in_prepass_ && !handover) {
      in_prepass_ = true;   // the same kernel over the first rows only
      Status ps = scan_gemm(ctx, d_q, nq, k, pre_rows, d_allow, allow_nbits, d_out_d, d_out_l, d_out_n, s);
      in_prepass_ = false;
      VK_TRY(ps);
      VK_TRY(ctx->d_stats.ensure(std::max<size_t>(32, nq * 8)));
      VK_HIP_TRY(launch_kth_bound(d_out_d, d_out_n, (uint32_t)k, (uint32_t)nq, ctx->d_stats.as<float>(), s));
      init_bound = ctx->d_stats.as<float>();
    }
    FlatGemmArgs g{};
    g.init_bound = init_bound;
    g.prepass = in_prepass_ ? 1 : 0;
    g.rows = store_.d_rows();
    g.bf16 = store_.bf16() ? 1 : 0;
    g.labels = store_.d_labels();
    g.queries = d_q;
    g.allow_bits = d_allow;
    g.allow_nbits = allow_nbits;
    g.row_stride_f = g.q_stride_f = store_.stride_f();
    g.chunks = store_.stride_f() / 16;
    g.n_rows = (uint32_t)count;
    g.nq = (uint32_t)nq;
    g.k = (uint32_t)k;
    g.tile_q = flat_gemm_tile_q(store_.stride_f());
    g.nqt = (uint32_t)((nq + g.tile_q - 1) / g.tile_q);
    const uint32_t tiles = (uint32_t)((count + 127) / 128);
    uint32_t nrp = std::max<uint32_t>(8, (256 / g.nqt) & ~7u);
    nrp = std::min<uint32_t>(nrp, std::max<uint32_t>(8, (tiles + 7) & ~7u));
    g.nrp = nrp;
    const uint64_t per_q = (uint64_t)nrp * 8 * k;
    VK_TRY(ctx->d_part_d.ensure((size_t)nq * per_q * 4));
    VK_TRY(ctx->d_part_l.ensure((size_t)nq * per_q * 8));
    g.part_dist = ctx->d_part_d.as<float>();
    g.part_label = ctx->d_part_l.as<uint64_t>();
    g.lockstep = g.nqt > 1 && g.nqt <= 32 && !handover ? gemm_lockstep_ : 0;
    g.contig = gemm_contig_;
    g.cancel = in_prepass_ ? nullptr : d_cancel;
    g.run_flag = run_flag;
    g.run_if = run_if;
    g.run_hi = run_hi;
    // [ progress words | per-query bounds ]
    const size_t sync_bytes = (size_t)nrp * 4 * 32 * 4;
    VK_TRY(ctx->d_sync.ensure(sync_bytes + nq * 4));
    g.sync = ctx->d_sync.as<uint32_t>();
    g.qbound = g.sync + sync_bytes / 4;
    if (!handover) {   // (the hand-over launch neither waits on the progress words nor reads the shared bound)
      VK_HIP_TRY(hipMemsetAsync(g.sync, 0, sync_bytes, s));
      if (init_bound)   // lists kept in HBM (k > 10) read the shared bound once per tile: start it at the pre-pass bound
        VK_HIP_TRY(hipMemcpyAsync(g.qbound, reinterpret_cast<const uint32_t *>(init_bound) + nq, nq * 4, hipMemcpyDeviceToDevice, s));
      else
        VK_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(g.qbound), (int)0xFF800000u, nq, s));   // key of +inf
    }
    VK_HIP_TRY(launch_flat_gemm(g, s));
    MergeArgs m{};
    m.in_dist = g.part_dist;
    m.in_label = g.part_label;
    m.part_stride = nq * per_q;
    m.q_stride = per_q;
    m.parts = 1;
    m.per_part = (uint32_t)per_q;
    m.k = (uint32_t)k;
    m.out_ld = (uint32_t)(in_prepass_ ? k : out_ld);
    m.out_dist = d_out_d;
    m.out_label = d_out_l;
    m.out_n = d_out_n;
    m.run_flag = run_flag;
    m.run_if = run_if;
    m.run_hi = run_hi;
    VK_HIP_TRY(launch_merge_topk(m, flat_scan_slots_per_lane(k), nq, s));
    return Status::Ok();
  }

  // a finished (t0, t1) pair around the filter launches of one batch -> the index's totals
  void drain_timed(SearchCtx::TimedPair &tp) {
    if (!tp.pending || hipEventQuery(tp.t1) != hipSuccess) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, tp.t0, tp.t1) == hipSuccess) {
      filter_ns_total_ += (uint64_t)((double)ms * 1e6);
      filter_timed_ += 1;
    }
    tp.pending = false;
  }

  // The sample behind the candidate filter's bound: every s-th 128-row tile of the index, about `filter_prepass_rows_`
  // rows per 10 of k and at most 1/32 of the index (so that the pass over it stays a few per cent of the batch).  The
  // final pass keeps about count * k / sample rows per query (380 at 10M rows, k = 10).
  uint64_t filter_prepass_rows(uint64_t k) const {
    const uint64_t kk = (k + 9) / 10, cap = filter_prepass_rows_ * kk;
    return std::max<uint64_t>(std::min<uint64_t>(cap, count_ / 32), std::min<uint64_t>(cap, 1024 * kk));
  }

  // Row statistics for the candidate filter -- per 128-row tile the largest row norm (or "not an f16 tile"), for L2 the
  // half norms -- brought up to date for the rows written since the last call (once after a writer phase; the searches
  // of a reader phase find nothing to do).  filter_bad_tiles_ = tiles the f16 pipe cannot carry.
  Status ensure_row_stats() {
    std::lock_guard<std::mutex> g(stats_mu_);
    uint64_t lo, hi;
    bool all = false;
    if (d_rowstats_.p == nullptr) {
      VK_TRY(d_rowstats_.ensure(64));
      VK_HIP_TRY(hipMemsetAsync(d_rowstats_.p, 0, 64, store_.stream()));
      all = true;
    }
    const size_t tile_bytes = (size_t)((store_.alloc_rows() + RowStore::kRowSlack + 127) / 128) * 4;
    if (d_tile_norm_.cap < tile_bytes) {   // (the per-tile tables follow the row table's size)
      VK_TRY(d_tile_norm_.ensure(tile_bytes));
      VK_HIP_TRY(hipMemsetAsync(d_tile_norm_.p, 0, d_tile_norm_.cap, store_.stream()));
      VK_HIP_TRY(hipMemsetAsync(d_rowstats_.p, 0, 64, store_.stream()));
      all = true;
    }
    if (l2() && d_hn16_.cap < store_.alloc_rows() * 4) {
      VK_TRY(d_hn16_.ensure(store_.alloc_rows() * 4));
      all = true;
    }
    if (store_.take_written(&lo, &hi) || all) {
      // The tile norms and the bad-tile count only ever grow (atomicMax): rows that were overwritten or removed leave their
      // old maxima behind -- valid bounds, but an index that once held long or non-finite rows would keep wide gates (or stay
      // off this path) for ever.  Once the rows rewritten since the last full pass add up to a quarter of the index, start over.
      rewritten_ += all ? 0 : hi - lo;
      if (!all && rewritten_ * 4 > std::max<uint64_t>(count_, 1024)) {
        VK_HIP_TRY(hipMemsetAsync(d_tile_norm_.p, 0, d_tile_norm_.cap, store_.stream()));
        VK_HIP_TRY(hipMemsetAsync(d_rowstats_.p, 0, 64, store_.stream()));
        all = true;
      }
      if (all) { lo = 0; hi = count_; rewritten_ = 0; }
      hi = std::min<uint64_t>(hi, store_.alloc_rows());
      Status st = [&]() -> Status {
        VK_HIP_TRY(launch_row_stats(store_.d_rows(), store_.bf16(), l2(), store_.stride_f(), (uint32_t)lo, (uint32_t)hi,
                                    (uint32_t)((count_ + 127) / 128), d_rowstats_.as<uint32_t>(), d_tile_norm_.as<uint32_t>(),
                                    l2() ? d_hn16_.as<uint32_t>() : nullptr, store_.stream()));
        uint32_t h[3] = {0, 0, 0};
        VK_HIP_TRY(hipMemcpyAsync(h, d_rowstats_.p, sizeof h, hipMemcpyDeviceToHost, store_.stream()));
        VK_HIP_TRY(hipStreamSynchronize(store_.stream()));
        filter_bad_tiles_.store(h[2], std::memory_order_relaxed);
        return Status::Ok();
      }();
      if (!st.ok()) {   // the range was consumed above: put it back, or the next search would gate with stale tile norms
        store_.note_written(lo, hi);
        return st;
      }
    }
    return Status::Ok();
  }

  // K4h pipeline (flat_filter.hip), six launches for a batch that needs no hand-over:
  //   qprep    queries -> f16 fragments, per-query error polynomials, counters cleared
  //   sample   the filter kernel over every s-th tile: per (64 rows, query) a lower bound of the group's best exact score
  //   select   per query the k-th largest group bound = a lower bound of its k-th best exact score
  //   filter   one pass over all rows: the pairs that can still be among a query's k best -> survivor lists
  //   re-rank  exact distances of the survivors (quad kernel in list mode)
  //   merge    (distance, label) selection; queries that lost survivors are listed for the redo
  // and behind them, returning at once unless the redo list is non-empty: the exact scan over the listed queries (up to
  // kFilterRedoMax), or the exact kernels over the whole batch (more).  Same answer as the exact path, bit for bit.
  Status scan_filter(SearchCtx *ctx, const float *d_q, uint64_t nq, uint64_t k, uint64_t count, const uint64_t *d_allow,
                     uint64_t allow_nbits, float *d_out_d, uint64_t *d_out_l, uint32_t *d_out_n, hipStream_t s,
                     uint64_t out_ld, const uint32_t *d_cancel) {
    const uint32_t dp = store_.stride_f();
    const uint32_t nqt = (uint32_t)((nq + 31) / 32);
    const uint32_t cap = (uint32_t)std::max<uint64_t>(filter_cap_, 64 * k);
    const int e = flat_scan_slots_per_lane(k);
    const uint32_t nrp = 8;
    const uint64_t per_q = (uint64_t)nrp * (e == 1 ? 1 : 4) * k;   // e == 1: one list per block (merged in LDS), else per wave
    // the sample: n_s sample tiles of 128 rows, row (i * n_s + t) * gap of the index being row i of sample tile t
    if (filter_blocks_ == 0) {
      int cus = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, store_.device());
      filter_blocks_ = cus > 0 ? (uint32_t)cus : 256;
    }
    // Two passes over the rows (option filter-two-pass; an index of at least filter-two-pass-min-tiles = 16 tiles per block): the sample only has to
    // keep the EARLY pass's survivor lists short -- it shrinks eightfold (213 us -> 30 us at 10M x 768) -- and the early pass,
    // the head of every block's tile range (256 stretches spread over the index, about sqrt(sample / rows) of it), is both
    // part of the scan and a sample sixteen times the old one: the main pass's bound is the k-th best of ITS survivors
    // (flat_bound_tighten_kernel).  Survivors per query ~ k (rows f / sample + 1 / f).
    const uint64_t all_tiles = (count + 127) / 128;
    const uint32_t filter_grid = (uint32_t)std::min<uint64_t>(filter_blocks_, all_tiles);
    const uint64_t tiles_per_block = all_tiles / filter_grid;
    uint64_t sample_rows = filter_prepass_rows(k);
    uint32_t early_tiles = 0;
    if (filter_two_pass_ != 0 && tiles_per_block >= filter_two_pass_min_tiles_ && !filter_dump_active()) {
      const uint64_t kk = (k + 9) / 10;
      sample_rows = std::min<uint64_t>(sample_rows, std::max<uint64_t>(8192 * kk, std::min<uint64_t>(32768 * kk, count / 256)));
      const uint64_t pm = opt_.get(kOptFilterEarlyPermille);
      const double f = pm != 0 ? (double)pm / 1000.0 : std::sqrt((double)sample_rows / (double)count);
      early_tiles = (uint32_t)std::max<double>(1.0, std::floor((double)tiles_per_block * f + 0.5));
      early_tiles = (uint32_t)std::min<uint64_t>(early_tiles, std::max<uint64_t>(1, tiles_per_block / 4));
    }
    uint64_t n_s = std::min<uint64_t>(std::max<uint64_t>(1, sample_rows / 128), (uint64_t)kFilterMaxGroups / 2);
    n_s = std::min<uint64_t>(n_s, count / 128);
    const uint32_t gap = (uint32_t)std::max<uint64_t>(1, count / (128 * n_s));
    // group bounds per query: two per sample tile (64 rows each), eight (16 rows each) while that stays within what the
    // selection holds at 16 values per thread -- the k-th largest of MANY narrow groups is a tighter bound than of few wide
    // ones, which is what a small sample needs
    const bool fine = n_s * 8 <= 4096;
    const uint32_t groups = (uint32_t)n_s * (fine ? 8 : 2);
    const uint32_t smax_ld = (groups + 63) & ~63u;
    const uint32_t n_chunks = filter_spill_chunks_;
    // per-batch words: [nq] survivor counts | [nq] hand-over flags | [4] spill_next, redo_cnt | [nq] redo list | [nq][32] chunk slots
    // ... | [nq] arrival counters of the fused re-rank
    const size_t w_cnt = 0, w_ovf = nq, w_misc = 2 * nq, w_redo = 2 * nq + 4, w_chunk = 3 * nq + 4, w_done = w_chunk + nq * kSpillPerQuery;
    VK_TRY(ctx->d_fcnt.ensure((w_done + 2 * nq) * 4));          // (... | [nq] rows the re-rank evaluated)
    VK_TRY(ctx->d_fq16.ensure((size_t)nqt * 32 * dp * 2));
    VK_TRY(ctx->d_fthr.ensure((size_t)nqt * 32 * (16 + 4 + 4)));      // error polynomials, then bounds, then witness margins
    VK_TRY(ctx->d_fcand.ensure(nq * (size_t)cap * 8));                 // row slots, then the scores
    VK_TRY(ctx->d_fspill.ensure((size_t)n_chunks * kSpillChunk * 8));
    VK_TRY(ctx->d_fsmax.ensure(nq * (size_t)smax_ld * 4));
    VK_TRY(ctx->d_fpart_d.ensure(nq * per_q * 4));
    VK_TRY(ctx->d_fpart_l.ensure(nq * per_q * 8));
    uint32_t *words = ctx->d_fcnt.as<uint32_t>();
    uint32_t *redo_cnt = words + w_misc + 1;
#ifdef VK_EXPERIMENTS
    // (the experiments build of the library only -- scripts/filter_ablate.py, scripts/build_experiments.sh: kernels whose
    //  ANSWERS ARE INVALID, selected through the environment per launch; none of this exists in libvkindex.so)
    const bool timing = getenv("VK_FILTER_TIMING") && atoi(getenv("VK_FILTER_TIMING")) != 0;
    const bool filter_experiment = getenv("VK_FILTER_TIMING") || (getenv("VK_FILTER_ABLATE") && !getenv("VK_FILTER_ABLATE_DMA"));
#else
    constexpr bool filter_experiment = false;
#endif

    FlatFilterArgs f{};
    f.rows = store_.d_rows();
    f.bf16 = store_.bf16() ? 1 : 0;
    f.l2 = l2() ? 1 : 0;
    // final-pass variants behind the same gate (options, A/B): bf16 rows on the bf16 matrix cores, rows / B operands by DMA
    f.qbf16 = (store_.bf16() && !l2() && opt_.get(kOptFilterBf16Mfma) != 0 && !filter_experiment) ? 1 : 0;
    f.dma = (f.qbf16 && opt_.get(kOptFilterRowDma) != 0) ? 1 : 0;
    f.bdma = opt_.get(kOptFilterBDma) != 0 ? 1u : 0u;
    f.hn16 = l2() ? d_hn16_.as<uint32_t>() : nullptr;
    f.labels = store_.d_labels();
    f.allow_bits = d_allow;
    f.allow_nbits = allow_nbits;
    f.queries = d_q;
    f.q_stride_f = dp;
    f.q16 = ctx->d_fq16.p;
    f.qcoef = ctx->d_fthr.as<float4>();
    f.qbound = reinterpret_cast<float *>(ctx->d_fthr.as<char>() + (size_t)nqt * 32 * 16);
    f.qwit = f.qbound + (size_t)nqt * 32;
    f.norm_cap = d_rowstats_.as<uint32_t>() + 3;
    f.sample_gap = gap;
    f.tile_norm = d_tile_norm_.as<uint32_t>();
    f.cand_cnt = words + w_cnt;
    f.cand_row = ctx->d_fcand.as<uint32_t>();
    f.cand_val = reinterpret_cast<float *>(f.cand_row + nq * (size_t)cap);
    f.cap = cap;
    f.qchunk = words + w_chunk;
    f.spill = ctx->d_fspill.as<uint32_t>();
    f.spill_val = reinterpret_cast<float *>(f.spill + (size_t)n_chunks * kSpillChunk);
    f.spill_next = words + w_misc;
    f.n_chunks = n_chunks;
    f.ovf_q = words + w_ovf;
    f.done_cnt = words + w_done;
    f.rerank_cnt = words + w_done + nq;
    f.smax = ctx->d_fsmax.as<float>();
    f.smax_ld = smax_ld;
    f.smax_fine = fine ? 1 : 0;
    f.row_stride_f = dp;
    f.n_rows = (uint32_t)count;
    f.nq = (uint32_t)nq;
    f.nqt = nqt;
    f.cancel = d_cancel;
    f.redo_cnt = redo_cnt;                   // (cleared by qprep, counted by the merge four launches later)
    VK_HIP_TRY(launch_flat_qprep(f, s));
    // one launch per 256 queries, every launch one pass over its tiles
    auto filter_launches = [&](FlatFilterArgs base, uint32_t blocks) -> Status {
      for (uint32_t g0 = 0; g0 < nqt; g0 += 8) {
        FlatFilterArgs fg = base;
        const size_t c0 = (size_t)g0 * 32;
        fg.nqt = std::min<uint32_t>(8, nqt - g0);
        fg.nq = (uint32_t)std::min<uint64_t>(256, nq - c0);
        fg.q16 = static_cast<char *>(base.q16) + c0 * dp * 2;
        fg.qcoef = base.qcoef + c0;
        fg.qbound = base.qbound + c0;
        fg.qwit = base.qwit + c0;
        fg.cand_cnt = base.cand_cnt + c0;
        fg.cand_row = base.cand_row + c0 * cap;
        fg.cand_val = base.cand_val + c0 * cap;
        fg.qchunk = base.qchunk + c0 * kSpillPerQuery;
        fg.ovf_q = base.ovf_q + c0;
        fg.done_cnt = base.done_cnt + c0;
        fg.rerank_cnt = base.rerank_cnt + c0;
        fg.smax = base.smax + c0 * smax_ld;
        VK_HIP_TRY(launch_flat_filter(fg, blocks, s));
      }
      return Status::Ok();
    };
    // 1. the bound: group bounds over the sample, then the k-th largest per query
    {
      FlatFilterArgs fs = f;
      fs.mode = 1;
      fs.n_tiles = (uint32_t)n_s;
      fs.cancel = nullptr;
      VK_TRY(filter_launches(fs, (uint32_t)std::min<uint64_t>(filter_blocks_, n_s)));
      FlatBoundArgs b{};
      b.smax = f.smax;
      b.smax_ld = smax_ld;
      b.groups = groups;
      b.k = (uint32_t)k;
      b.nq = (uint32_t)nq;
      b.qbound = f.qbound;
      VK_HIP_TRY(launch_flat_bound_select(b, s));
    }
    // 2. the filter over all rows
    {
      FlatFilterArgs fm = f;
      fm.mode = 0;
      fm.n_tiles = (uint32_t)((count + 127) / 128);
#ifdef VK_EXPERIMENTS
      fm.timing = timing && !l2();
      fm.prio = getenv("VK_FILTER_PRIO") ? (uint32_t)atoi(getenv("VK_FILTER_PRIO")) : 0u;
      if (!l2() && getenv("VK_FILTER_ABLATE") && (!fm.qbf16 || fm.dma)) {
        fm.ablate_on = 1;
        fm.ablate = (uint32_t)atoi(getenv("VK_FILTER_ABLATE"));
      }
      fm.dump_scores = g_exp_dump.scores;   // margin audit (vk_exp_filter_dump)
      fm.dump_thr = g_exp_dump.thr;
      fm.dump_rows = g_exp_dump.rows;
      fm.dump_ld = g_exp_dump.ld;
      if (fm.timing) {   // phase timing experiment: nine counters
        VK_TRY(ctx->d_idx.ensure(128));
        VK_HIP_TRY(hipMemsetAsync(ctx->d_idx.p, 0, 128, s));
        fm.dbg = ctx->d_idx.as<unsigned long long>();
      }
#endif
      filter_batches_.fetch_add(1, std::memory_order_relaxed);
      // kernel-level timing of the final pass (option kernel-timing, off by default: bench.py's roofline figure)
      SearchCtx::TimedPair *tp = nullptr;
      if (opt_.get(kOptKernelTiming) != 0) {
        tp = &ctx->timed[ctx->timed_next++ % 32];
        drain_timed(*tp);
        if (!tp->t0) {
          VK_HIP_TRY(hipEventCreate(&tp->t0));
          VK_HIP_TRY(hipEventCreate(&tp->t1));
        }
      }
      // (an event record is a marker the stream executes: 6 us of idle device each, seen as gaps in profiles/r05_step_trace_*.log
      //  -- ONE in front of the timed launch and one behind it, only with the option on)
      const bool two_pass = early_tiles != 0 && !filter_experiment;
      if (two_pass) {
        // the early pass (before the timed pair: kernel-timing brackets the main pass, the launch bench.py prices) ...
        FlatFilterArgs fe = fm;
        fe.part_first = 0;
        fe.part_tiles = early_tiles;
        fe.early = 1;
        VK_TRY(filter_launches(fe, filter_grid));
        // ... its survivors raise the bounds ...
        FlatTightenArgs t{};
        t.cand_cnt = f.cand_cnt;
        t.cand_row = f.cand_row;
        t.cand_val = f.cand_val;
        t.cap = cap;
        t.k = (uint32_t)k;
        t.nq = (uint32_t)nq;
        t.l2 = l2() ? 1u : 0u;
        t.qcoef = f.qcoef;
        t.tile_norm = f.tile_norm;
        t.qbound = f.qbound;
        VK_HIP_TRY(launch_flat_bound_tighten(t, s));
        // ... and the main pass walks the rest of every block's range
        fm.part_first = early_tiles;
        fm.part_tiles = 0xFFFFFFFFu;
      }
      if (tp) VK_HIP_TRY(hipEventRecord(tp->t0, s));
      last_filter_final_rows_.store(count - (two_pass ? (uint64_t)early_tiles * filter_grid * 128 : 0), std::memory_order_relaxed);
      VK_TRY(filter_launches(fm, filter_grid));
      if (tp) {
        VK_HIP_TRY(hipEventRecord(tp->t1, s));
        tp->pending = true;
      }
#ifdef VK_EXPERIMENTS
      if (g_exp_dump.qstate != nullptr && nq <= g_exp_dump.ld) {   // margin audit: the per-query gate state next to the scores
        for (int c = 0; c < 4; ++c)
          VK_HIP_TRY(hipMemcpy2DAsync(g_exp_dump.qstate + (size_t)c * g_exp_dump.ld, 4, reinterpret_cast<const float *>(f.qcoef) + c, 16, 4, nq,
                                      hipMemcpyDeviceToDevice, s));
        VK_HIP_TRY(hipMemcpyAsync(g_exp_dump.qstate + (size_t)4 * g_exp_dump.ld, f.qbound, nq * 4, hipMemcpyDeviceToDevice, s));
        VK_HIP_TRY(hipMemcpyAsync(g_exp_dump.qstate + (size_t)5 * g_exp_dump.ld, f.qwit, nq * 4, hipMemcpyDeviceToDevice, s));
      }
      if (fm.timing) {
        unsigned long long h[9];
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(filter_blocks_, fm.n_tiles);
        VK_HIP_TRY(hipStreamSynchronize(s));
        VK_HIP_TRY(hipMemcpy(h, ctx->d_idx.p, sizeof h, hipMemcpyDeviceToHost));
        const double w2 = (double)blocks * 2, w4 = (double)blocks * 4;
        fprintf(stderr, "[vk] filter phases, cycles per wave: row producers issue %.0f  wait+convert+store %.0f  barrier %.0f | "
                        "query producers issue+wait+store %.0f  barrier %.0f | consumers mfma %.0f  gate %.0f  barrier %.0f\n",
                h[0] / w2, h[2] / w2, h[3] / w2, h[4] / w2, h[8] / w2, h[5] / w4, h[6] / w4, h[7] / w4);
      }
#endif
    }
    // 3. exact re-rank of the survivors + selection
    {
      FlatScanArgs r{};
      r.rows = store_.d_rows();
      r.labels = store_.d_labels();
      r.queries = d_q;
      r.allow_bits = d_allow;
      r.allow_nbits = allow_nbits;
      r.part_dist = ctx->d_fpart_d.as<float>();
      r.part_label = ctx->d_fpart_l.as<uint64_t>();
      r.row_stride_f = r.q_stride_f = dp;
      r.chunks = dp / 16;
      r.row_begin = 0;
      r.row_end = (uint32_t)count;
      r.nq = (uint32_t)nq;
      r.k = (uint32_t)k;
      r.nrp = nrp;
      r.nqg = (uint32_t)nq;
      r.cand_cnt = f.cand_cnt;
      r.cand_row = f.cand_row;
      r.cand_cap = cap;
      r.cand_qchunk = f.qchunk;
      r.cand_spill = f.spill;
      r.cand_ovf = f.ovf_q;
      const bool fused = e == 1 && opt_.get(kOptFlatFusedRerank) != 0;   // k <= 64: one launch writes the answers
      if (!fused) VK_HIP_TRY(launch_flat_scan(r, l2(), store_.bf16(), 1, e, s));
      MergeArgs m{};
      m.in_dist = r.part_dist;
      m.in_label = r.part_label;
      m.part_stride = nq * per_q;
      m.q_stride = per_q;
      m.parts = 1;
      m.per_part = (uint32_t)per_q;
      m.k = (uint32_t)k;
      m.out_ld = (uint32_t)out_ld;
      m.out_dist = d_out_d;
      m.out_label = d_out_l;
      m.out_n = d_out_n;
      m.ovf_q = f.ovf_q;
      m.redo_cnt = redo_cnt;
      m.redo_list = words + w_redo;
      if (fused) {
        r.cancel = d_cancel;
        r.done_cnt = f.done_cnt;
        r.reranked = f.rerank_cnt;
#ifdef VK_EXPERIMENTS
        r.stamps = g_exp_rerank_stamps;
#endif
        if (opt_.get(kOptFilterSecondBound) != 0) {   // the second bound: from the survivors' own approximate scores
          r.cand_val = f.cand_val;
          r.cand_spill_val = f.spill_val;
          r.qcoef = f.qcoef;
          r.tile_norm = f.tile_norm;
        }
        VK_HIP_TRY(launch_flat_rerank(r, m, l2(), store_.bf16(), s));
      } else {
        VK_HIP_TRY(launch_merge_topk(m, e, nq, s));
      }
    }
    // 4. the hand-over (decided on the device; these launches return at once when the redo list is empty): the exact
    //    scan over the listed queries alone, or -- more than kFilterRedoMax of them -- the exact kernels over the batch
    VK_TRY(scan_k3(ctx, d_q, kFilterRedoMax, k, count, d_allow, allow_nbits, d_cancel, d_out_d, d_out_l, d_out_n, s, out_ld, redo_cnt, 1,
                   kFilterRedoMax, words + w_redo));
    if (!l2() && k <= 10 && flat_gemm_supported(store_.stride_f(), k))
      VK_TRY(scan_gemm(ctx, d_q, nq, k, count, d_allow, allow_nbits, d_out_d, d_out_l, d_out_n, s, out_ld, d_cancel, nullptr, redo_cnt,
                       kFilterRedoMax + 1, 0xFFFFFFFFu));
    else
      VK_TRY(scan_k3(ctx, d_q, nq, k, count, d_allow, allow_nbits, d_cancel, d_out_d, d_out_l, d_out_n, s, out_ld, redo_cnt,
                     kFilterRedoMax + 1, 0xFFFFFFFFu));
    filter_used_ = true;
    return Status::Ok();
  }

  // candidate filter (K4h): switches and sizes -- run-time options (options.hpp), read with relaxed loads on the search path
  OptRef filter_enabled_{&opt_, kOptFlatFilter};
  OptRef filter_min_queries_{&opt_, kOptFilterMinQueries};
  OptRef filter_min_rows_{&opt_, kOptFilterMinRows};
  OptRef filter_prepass_rows_{&opt_, kOptFilterPrepassRows};
  OptRef filter_two_pass_{&opt_, kOptFilterTwoPass};
  OptRef filter_two_pass_min_tiles_{&opt_, kOptFilterTwoPassMinTiles};
  OptRef filter_cap_{&opt_, kOptFilterCap};
  // spill chunks (of kSpillChunk survivors) a batch's queries share beyond their private lists: 16 MB per context
  OptRef filter_spill_chunks_{&opt_, kOptFilterSpillChunks};
  uint32_t filter_blocks_ = 0;
  DevBuf d_rowstats_, d_hn16_, d_tile_norm_;
  std::atomic<uint32_t> filter_bad_tiles_{0};   // tiles the f16 pipe cannot carry (row_stats_kernel)
  std::mutex stats_mu_;
  uint64_t rewritten_ = 0;                      // rows brought up to date since the last full pass (under stats_mu_)
  uint64_t max_label_ = 0;   // the largest label ever held (vk_index_stats.max_label)
  std::atomic<uint64_t> last_filter_final_rows_{0};   // rows the main pass of the most recent batch walked
  std::atomic<uint64_t> last_filter_cands_{0}, last_filter_fallback_{0}, last_filter_reranked_{0}, filter_ns_total_{0}, filter_batches_{0}, filter_timed_{0};
  static thread_local bool filter_used_;
  static constexpr uint64_t kGemmMinQueries = 5;    // measured at 10Mx768: K3 4 queries 5.3 ms, 8 queries 11.7 ms; K4 up to 32 queries 6.1 ms
  static constexpr uint64_t kMaxPassK = 1024;
  // per-call lower bounds of search_in_passes (set only around its scan() calls, under the ctx lease)
  static thread_local const float *lb_dist_;
  static thread_local const uint64_t *lb_label_;
  RowStore store_;
  CtxPool pool_;
  std::shared_mutex rw_;
  // K4 lockstep window in row tiles (see FlatGemmArgs::lockstep); 0 turns it off
  OptRef gemm_lockstep_{&opt_, kOptGemmLockstep};
  // floor of the row partitions of a K3 launch (blocks = partitions x query groups)
  OptRef scan_min_nrp_{&opt_, kOptScanMinNrp};
  OptRef gemm_prepass_rows_{&opt_, kOptGemmPrepassRows};
  static thread_local bool in_prepass_;
  OptRef gemm_contig_{&opt_, kOptGemmContig};
  OptRef force_scan_{&opt_, kOptFlatForceScan};   // A/B switch for benchmarks: VALU scan for every batch size
  std::unordered_map<uint64_t, uint32_t> slot_of_;  // dict_external_to_internal
  uint64_t count_ = 0;                               // cur_element_count_
  uint64_t capacity_;                                // data_->getCapacity()
};

thread_local bool FlatIndex::in_prepass_ = false;
thread_local bool FlatIndex::filter_used_ = false;
thread_local const float *FlatIndex::lb_dist_ = nullptr;
thread_local const uint64_t *FlatIndex::lb_label_ = nullptr;

// ---- persistence: bruteforce.h:147-207 --------------------------------------------------------
// chunk 0: BruteForceIndexHeader{max_elements=1, size_per_element=2, curr_element_count=3}
// (index.proto), then one chunk per element: [vector bytes | label u64].
Status FlatIndex::save(vk_write_chunk_fn fn, void *user) {
  VK_TRY(flush_if_dirty());
  std::shared_lock<std::shared_mutex> lk(rw_);
  (void)hipSetDevice(store_.device());
  const size_t vec_bytes = (size_t)params_.dim * 4;
  std::string hdr;
  pb_put_varint_field(hdr, 1, capacity_);
  pb_put_varint_field(hdr, 2, vec_bytes + 8);
  pb_put_varint_field(hdr, 3, count_);
  if (fn(user, hdr.data(), hdr.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
  const size_t rb = store_.row_bytes();
  const uint64_t batch = std::max<uint64_t>(1, ((size_t)64 << 20) / rb);
  std::vector<char> rows(batch * rb), buf(vec_bytes + 8);
  for (uint64_t i0 = 0; i0 < count_; i0 += batch) {
    uint64_t nb = std::min<uint64_t>(batch, count_ - i0);
    VK_HIP_TRY(hipMemcpy(rows.data(), reinterpret_cast<const char *>(store_.d_rows()) + i0 * rb, nb * rb,
                         hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < nb; ++i) {
      if (store_.bf16()) {   // the stream carries f32 vectors (the widened, already rounded values)
        const uint16_t *h = reinterpret_cast<const uint16_t *>(rows.data() + i * rb);
        for (uint32_t d = 0; d < params_.dim; ++d) {
          uint32_t u = (uint32_t)h[d] << 16;
          memcpy(buf.data() + (size_t)d * 4, &u, 4);
        }
      } else
      memcpy(buf.data(), rows.data() + i * rb, vec_bytes);
      uint64_t lab = store_.host_labels()[i0 + i];
      memcpy(buf.data() + vec_bytes, &lab, 8);
      if (fn(user, buf.data(), buf.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
    }
  }
  return Status::Ok();
}

Status FlatIndex::load_from(vk_read_chunk_fn fn, void *user) {
  std::vector<char> buf((size_t)params_.dim * 4 + 64);
  uint64_t len = 0;
  if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
  uint64_t max_elements = 0, size_per_element = 0, cur = 0;
  PbReader r{reinterpret_cast<const uint8_t *>(buf.data()), reinterpret_cast<const uint8_t *>(buf.data()) + len};
  uint32_t field, wire;
  uint64_t val;
  while (r.next(&field, &wire, &val)) {
    if (field == 1) max_elements = val;
    else if (field == 2) size_per_element = val;
    else if (field == 3) cur = val;
  }
  if (size_per_element != (uint64_t)params_.dim * 4 + 8)
    return Status::Err(VK_ERR_INTERNAL, "Persisted size_per_element does not match expectation.");
  if (cur > max_elements) return Status::Err(VK_ERR_INTERNAL, "corrupt header: count exceeds max_elements");
  std::unique_lock<std::shared_mutex> lk(rw_);
  capacity_ = max_elements;
  for (uint64_t i = 0; i < cur; ++i) {
    if (fn(user, buf.data(), buf.size(), &len) || len != size_per_element)
      return Status::Err(VK_ERR_INTERNAL, "truncated element chunk");
    uint64_t lab;
    memcpy(&lab, buf.data() + (size_t)params_.dim * 4, 8);
    if (lab == ~0ull) return Status::Err(VK_ERR_INVALID, "label UINT64_MAX is reserved (the padding of result lists)");
    VK_TRY(observe_loaded_row(lab, buf.data()));
    VK_TRY(add_locked(lab, reinterpret_cast<const float *>(buf.data())));
    if (store_.staged_bytes() >= ((size_t)256 << 20)) VK_TRY(store_.flush());
  }
  return store_.flush();
}

static Status pick_device(const vk_index_params &p, int *device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return Status::Err(VK_ERR_NO_DEVICE, "no HIP device: libvkindex needs a gfx950 GPU (no CPU fallback)");
  int d = p.device_id;
  if (d < 0) {
    if (hipGetDevice(&d) != hipSuccess) d = 0;
  }
  if (d >= n) return Status::Err(VK_ERR_INVALID, "device_id out of range");
  *device = d;
  return Status::Ok();
}

Status create_flat(const vk_index_params &p, std::unique_ptr<Index> *out) {
  int device = 0;
  VK_TRY(pick_device(p, &device));
  *out = std::make_unique<FlatIndex>(p, device);
  return Status::Ok();
}

Status load_flat(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out) {
  int device = 0;
  VK_TRY(pick_device(p, &device));
  auto ix = std::make_unique<FlatIndex>(p, device);
  VK_TRY(ix->load_from(fn, user));
  *out = std::move(ix);
  return Status::Ok();
}

}  // namespace vk
