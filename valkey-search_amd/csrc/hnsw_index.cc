// hnsw_index.cc -- HnswIndex: hnswlib::HierarchicalNSW<float> semantics
// (third_party/hnswlib/hnswalg.h) with the graph built on the host (hnsw_graph.cc) and
// searched on the device (hnsw_search.hip).
//
//   addPoint / updatePoint / markDelete / resizeIndex -> HnswGraph (host, thread safe)
//   searchKnn (hnswalg.h:1659-1725)                   -> one wave per query on the GPU
//   SaveIndex / LoadIndex (hnswalg.h:808-1139)        -> same chunk stream, same load checks
// Device mirror: the row table (row_store.hpp), the level-0 table links0[cap][2M+1] (the
// reference's level-0 record without its pointer/label tail), and the upper-level lists
// packed in a pool of (M+1)-word slots addressed by upper_slot[id] + level - 1.  Host
// changes are tracked per node and published by flush().
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "hnsw_graph.hpp"
#include "index.hpp"

namespace vk {

// CPUs this process may actually use: hardware threads capped by the cgroup CPU quota
// (a container can expose 256 hardware threads with a 16-CPU quota; oversubscribing the quota
// throttles lock holders and collapses the build rate).
static unsigned effective_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  auto apply = [&](double quota, double period) {
    if (quota > 0 && period > 0) n = (unsigned)std::max(1.0, std::min((double)n, floor(quota / period + 0.5)));
  };
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota|max> <period>"
    char q[32];
    double period = 0;
    if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0) apply(atof(q), period);
    fclose(f);
  } else {
    double quota = -1, period = 0;                                // cgroup v1
    if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lf", &quota) != 1) quota = -1; fclose(fq); }
    if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lf", &period) != 1) period = 0; fclose(fp); }
    apply(quota, period);
  }
  return n;
}

class HnswIndex final : public Index {
 public:
  HnswIndex(const vk_index_params &p, int device)
      : Index(p), store_(device, p.dim, p.dtype == VK_DTYPE_BF16), pool_(device),
        graph_(std::make_unique<HnswGraph>(p.dim, p.metric == VK_METRIC_L2, p.initial_cap, p.m, p.ef_construction,
                                           p.random_seed, p.allow_replace_deleted != 0)) {
    graph_->set_ef(p.ef_runtime ? p.ef_runtime : 10);
  }
  ~HnswIndex() override {
    (void)hipSetDevice(store_.device());
    d_links0_.release();
    d_totals_.release();
    d_upper_slot_.release();
    d_upper_pool_.release();
  }

  // addPoint, one element (hnswalg.h:1278-1340).  IndexSchema feeds AddRecord one key at a time from the writer pool
  // (src/index_schema.cc:755-791), so this -- not add_batch -- is what a backfill or an ingest burst arrives through.  A NEW
  // label is therefore STAGED (its row copied, nothing linked) once the graph is large enough for the device build; what a
  // writer phase staged is linked in bulk by drain_pending() at vk_index_flush / before the next search -- on the device
  // (K9, hnsw_build.hip) when there are thousands, by the host threads otherwise.  The points of one bulk do not see one
  // another while they are linked: the relaxation concurrent addPoint calls already have (hnswalg.h:1523-1650 take no lock
  // across points).  Updates of an existing label, replace-deleted inserts and small graphs take the host builder at once.
  Status add(uint64_t label, const float *row) override {
    // Staging does not take the index lock: it touches the staging area alone (its own mutex), so the writers keep staging
    // while a bulk of earlier rows is being linked on the device under the exclusive lock.
    // A label that is staged, or in the bulk being linked right now, is looked for there whatever stage_candidate() says
    // by now (a remove of a linked element switches staging off under hnsw-allow-replace-deleted while rows still wait).
    const bool can_stage = stage_candidate();
    for (;;) {
      bool full = false, staged = false, in_bulk = false;
      {
        std::lock_guard<std::mutex> pl(pend_.mu);
        auto it = pend_.pos.find(label);
        uint32_t id;
        if (it != pend_.pos.end()) {   // the same label again before it was linked: the later row wins (an in-place update)
          memcpy(pend_.rows.data() + it->second * params_.dim, row, (size_t)params_.dim * 4);
          return Status::Ok();
        }
        if (pend_.linking.count(label)) {
          in_bulk = true;   // its first row is being linked right now: this one is an update behind that bulk
        } else if (can_stage && !graph_->lookup(label, &id)) {   // (an update of a linked element goes to the host builder below)
          if (graph_->count() + pend_.live + draining_.load(std::memory_order_relaxed) >= graph_->max_elements())
            return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
          if (pend_.rows.capacity() == 0) pend_.rows.reserve((size_t)std::min<uint64_t>(opt_.get(kOptHnswStageMax), 1u << 20) * params_.dim);
          pend_.pos.emplace(label, pend_.labels.size());
          pend_.labels.push_back(label);
          pend_.rows.insert(pend_.rows.end(), row, row + params_.dim);
          pend_.live += 1;
          graph_->note_label(label);
          staged_adds_.fetch_add(1, std::memory_order_relaxed);
          full = pend_.labels.size() >= opt_.get(kOptHnswStageMax);
          staged = true;
        }
      }
      // (the staging area is bounded: a writer that finds it full links what is there -- or waits for the writer that is
      //  doing so -- before it returns, like a caller of one long add_batch)
      if (staged) return full ? drain_pending(/*only_if_full=*/true) : Status::Ok();
      if (!in_bulk) break;
      wait_for_bulk();   // ... and look again: a bulk that FAILED put the rows that had not reached the graph back into staging
    }
    std::shared_lock<std::shared_mutex> lk(rw_);
    return add_one(label, row);
  }

  Status add_batch(const uint64_t *labels, const float *rows, uint64_t n) override {
    VK_TRY(drain_pending());   // (order of effects = order of calls)
    return add_batch_now(labels, rows, n, nullptr);
  }

  Status add_batch_now(const uint64_t *labels, const float *rows, uint64_t n, bool *on_device) {
    if (on_device) *on_device = false;
    if (device_build_ && n >= kDeviceBuildMinBatch) {
      bool handled = false;
      Status s = add_batch_device(labels, rows, n, &handled);
      if (on_device) *on_device = handled;
      if (handled || !s.ok()) return s;
    }
    return add_batch_host(labels, rows, n);
  }

  // link what the single adds staged (see add()); called without the index lock.  only_if_full: the caller is a writer that
  // found the staging area full -- when it gets its turn behind the writer that linked that bulk and fewer rows wait than a
  // device bulk takes, it goes back to staging (sixteen writers queue up behind one bulk; each of them linking the handful of
  // rows the others staged meanwhile sent up to 3 700 of a million rows to the host builder in bulks of tens:
  // scripts/staged_remainder_probe.py); with a device bulk's worth waiting it links them at once, the device does not idle.
  Status drain_pending(bool only_if_full = false) {
    {   // (cheap exit: nothing staged and nobody linking -- a flush, a save or a search that arrives while ANOTHER thread
        //  links what it swapped out waits for it below: every add that was acknowledged is in the graph when this returns)
      std::lock_guard<std::mutex> pl(pend_.mu);
      if (pend_.labels.empty() && draining_.load(std::memory_order_relaxed) == 0) return Status::Ok();
    }
    std::lock_guard<std::mutex> one_at_a_time(drain_mu_);
    std::vector<float> &rows = spare_rows_;       // (the buffers keep their capacity from bulk to bulk: 800 MB at the default
    std::vector<uint64_t> &labels = spare_labels_;   //  stage size is not reallocated and re-faulted every time)
    rows.clear();
    labels.clear();
    uint64_t waiting = 0;
    {
      std::lock_guard<std::mutex> pl(pend_.mu);
      if (pend_.labels.empty()) return Status::Ok();
      if (only_if_full && pend_.labels.size() < std::min<uint64_t>(opt_.get(kOptHnswStageMax), kDeviceBuildMinBatch)) return Status::Ok();
      waiting = pend_.labels.size();
    }
    // The row table grows BEFORE the bulk leaves the staging area: if the device has no room for it, the call fails here with
    // every acknowledged add still staged and nothing registered in the graph (the usual way to run out: the table of a large
    // index growing by half).  What can still fail behind this point is a few megabytes of scratch; see the end of this function.
    {
      bool grow;
      {
        std::shared_lock<std::shared_mutex> lk(rw_);
        grow = graph_->count() + waiting > store_.alloc_rows();
      }
      if (grow) {
        std::unique_lock<std::shared_mutex> lk(rw_);
        (void)hipSetDevice(store_.device());
        VK_TRY(store_.reserve(graph_->count() + waiting));
      }
    }
    {
      std::lock_guard<std::mutex> pl(pend_.mu);
      if (pend_.labels.empty()) return Status::Ok();
      if (pend_.live == pend_.labels.size()) {
        rows.swap(pend_.rows);
        labels.swap(pend_.labels);
      } else {   // some were removed again before they were linked
        const uint32_t dim = params_.dim;
        for (size_t i = 0; i < pend_.labels.size(); ++i) {
          auto it = pend_.pos.find(pend_.labels[i]);
          if (it == pend_.pos.end() || it->second != i) continue;
          labels.push_back(pend_.labels[i]);
          rows.insert(rows.end(), pend_.rows.begin() + i * dim, pend_.rows.begin() + (i + 1) * dim);
        }
        pend_.rows.clear();
        pend_.labels.clear();
      }
      // The labels of the bulk stay findable until they are in the graph: between this swap and the moment the builder has
      // registered them a remove / contains / get_row / add of one of them would otherwise find it nowhere (a delete lost, a
      // ghost vector linked behind it).  The map itself moves over -- no per-label work.
      pend_.linking.clear();
      pend_.linking.swap(pend_.pos);
      pend_.live = 0;
      draining_.store(labels.size(), std::memory_order_relaxed);
    }
    struct Done {
      Pending &p; std::atomic<uint64_t> &d;
      ~Done() { std::lock_guard<std::mutex> pl(p.mu); p.linking.clear(); d.store(0, std::memory_order_relaxed); }
    } done{pend_, draining_};
    if (labels.empty()) return Status::Ok();
    bool on_device = false;
    Status st = add_batch_now(labels.data(), rows.data(), labels.size(), &on_device);
    if (st.ok() && on_device) staged_adds_device_.fetch_add(labels.size(), std::memory_order_relaxed);
    if (!st.ok()) restage_after_failure(labels, rows);
    return st;
  }

  // A bulk failed part of the way (the error goes to the caller of the drain): no acknowledged add may be lost with it.
  // Rows that never reached the graph go back into the staging area (remove / get_row / a second add of such a label look
  // there again after waiting for the bulk); elements that were registered for a batch the device never linked -- in the
  // graph, with an empty level-0 list, reachable by nothing -- are linked by the host builder's update path, which needs
  // the host graph alone.  Called with drain_mu_ held, `linking` still naming the bulk.
  void restage_after_failure(const std::vector<uint64_t> &labels, const std::vector<float> &rows) {
    const uint32_t dim = params_.dim;
    std::vector<size_t> back;
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      for (size_t i = 0; i < labels.size(); ++i) {
        uint32_t id;
        if (!graph_->lookup(labels[i], &id)) back.push_back(i);
        else if ((graph_->links0(id)[0] & 0xFFFFu) == 0 && !graph_->is_deleted(id) && graph_->count() > 1)
          (void)add_one(labels[i], rows.data() + i * dim);
      }
    }
    std::lock_guard<std::mutex> pl(pend_.mu);
    for (size_t i : back) {
      if (!pend_.pos.emplace(labels[i], pend_.labels.size()).second) continue;
      pend_.labels.push_back(labels[i]);
      pend_.rows.insert(pend_.rows.end(), rows.begin() + i * dim, rows.begin() + (i + 1) * dim);
      pend_.live += 1;
    }
  }

  Status add_batch_host(const uint64_t *labels, const float *rows, uint64_t n) {
    // The rows of a batch take effect in their order.
    // (1) A batch that does not fit is cut where addPoint would throw: everything in front of that row is in, nothing behind
    //     it (the builder's threads would otherwise stop wherever they were when one of them met the limit: rows in front of
    //     the failing one left out, rows behind it in).
    // (2) A label twice in what is left means the later row stands: the threads take rows as they come, so only the LAST row
    //     of such a label is handed to them (r06: with three rows of one label in a batch the first one could win).
    std::shared_lock<std::shared_mutex> lk(rw_);
    Status over = Status::Ok();
    std::vector<uint64_t> ul;
    std::vector<float> ur;
    if (n > 1 || (n == 1 && graph_->count() >= graph_->max_elements())) {
      uint64_t room = graph_->max_elements() > graph_->count() ? graph_->max_elements() - graph_->count() : 0;
      if (params_.allow_replace_deleted) room += graph_->deleted_count();   // (such an insert takes over a tombstoned slot)
      std::unordered_map<uint64_t, uint64_t> last;      // label -> its last row in front of the cut
      std::unordered_set<uint64_t> fresh;               // new labels in front of the cut
      last.reserve(n * 2);
      for (uint64_t i = 0; i < n; ++i) {
        const uint64_t label = labels ? labels[i] : i;
        uint32_t id;
        if (!graph_->lookup(label, &id) && !fresh.count(label)) {
          if (fresh.size() == room) {
            n = i;
            over = Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
            break;
          }
          fresh.insert(label);
        }
        last[label] = i;
      }
      if (last.size() < n) {
        const uint32_t dim = params_.dim;
        ul.reserve(last.size());
        ur.reserve(last.size() * dim);
        for (uint64_t i = 0; i < n; ++i) {
          const uint64_t label = labels ? labels[i] : i;
          if (last[label] != i) continue;
          ul.push_back(label);
          ur.insert(ur.end(), rows + i * dim, rows + (i + 1) * dim);
        }
        labels = ul.data();
        rows = ur.data();
        n = ul.size();
      }
    }
    unsigned threads = params_.build_threads ? params_.build_threads : effective_cpus();
    threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, n / 16 + 1));
    if (threads == 1) {
      for (uint64_t i = 0; i < n; ++i) VK_TRY(add_one(labels ? labels[i] : i, rows + i * params_.dim));
      return over;
    }
    // the first element is inserted alone so every worker starts from a non-empty graph
    uint64_t first = 0;
    if (graph_->count() == 0 && n) {
      VK_TRY(add_one(labels ? labels[0] : 0, rows));
      first = 1;
    }
    std::atomic<uint64_t> next{first};
    std::atomic<bool> failed{false};
    Status err;
    std::mutex err_mu;
    auto work = [&]() {
      for (;;) {
        uint64_t i = next.fetch_add(1);
        if (i >= n || failed.load()) return;
        Status s = add_one(labels ? labels[i] : i, rows + i * params_.dim);
        if (!s.ok()) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!failed.exchange(true)) err = s;
          return;
        }
      }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto &t : pool) t.join();
    return failed.load() ? err : over;
  }

  Status remove(uint64_t label) override {
    for (;;) {
      bool in_bulk = false;
      {
        std::lock_guard<std::mutex> pl(pend_.mu);
        auto it = pend_.pos.find(label);
        if (it != pend_.pos.end()) {   // staged and not linked yet: it never enters the graph
          pend_.pos.erase(it);
          pend_.live -= 1;
          return Status::Ok();
        }
        in_bulk = pend_.linking.count(label) != 0;
      }
      if (!in_bulk) break;
      wait_for_bulk();   // (being linked right now: the tombstone goes on the linked element -- or, the bulk having failed, the
                         //  row is back in the staging area: look again)
    }
    std::shared_lock<std::shared_mutex> lk(rw_);
    return graph_->mark_delete(label);
  }

  // The bulk that drain_pending() swapped out is in the graph when this returns.  Called WITHOUT the index lock (the device
  // build takes it exclusively).
  void wait_for_bulk() { std::lock_guard<std::mutex> behind_the_bulk(drain_mu_); }

  Status resize(uint64_t new_max) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    return graph_->resize(new_max);
  }

  Status set_ef(uint32_t ef) override {
    graph_->set_ef(ef);
    return Status::Ok();
  }

  // (nothing staged or dirty: no exclusive lock, so a flush at every write -> read switch does not wait for the batches in flight)
  Status flush() override { return flush_if_dirty(); }

  Status search(const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) override {
    VK_TRY(flush_if_dirty());
    std::shared_lock<std::shared_mutex> lk(rw_);
    (void)hipSetDevice(store_.device());
    if (rq.nq == 0) return Status::Ok();
    if (pub_.count == 0 || rq.k == 0) {
      for (uint64_t q = 0; q < rq.nq; ++q) out_n[q] = 0;
      return Status::Ok();
    }
    if (cancel_raised(rq.cancel_flag) && !rq.partial_ok)
      return Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
    CtxLease lease(pool_);
    SearchCtx *ctx = lease.ctx;
    VK_TRY(upload_queries(ctx, rq.queries, rq.nq, params_.dim, store_.stride_f(), opt_.get(kOptUploadParallel) != 0, rq.query_tab));
    const uint64_t *d_allow = nullptr;
    uint64_t allow_nbits = rq.allow_nbits;
    if (rq.filter) {   // a device-resident filter for the whole batch: nothing to upload
      d_allow = rq.filter->bits_on(store_.device());
      allow_nbits = rq.filter->nbits();
      if (!d_allow) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's device");
    } else {
      VK_TRY(upload_allow(ctx, rq.allow_bits, rq.allow_nbits, &d_allow));
    }
    const uint64_t *const *d_tab = nullptr;
    const uint64_t *d_tab_nbits = nullptr;
    VK_TRY(build_filter_table(ctx, rq, ctx->stream, &d_tab, &d_tab_nbits));
    tab_ = d_tab;
    tab_nbits_ = d_tab_nbits;
    // (launch() consumes and clears them; an early return before it must not leave this batch's table behind for the
    // thread's next launch, e.g. a search_device or a device build)
    struct TabReset { ~TabReset() { tab_ = nullptr; tab_nbits_ = nullptr; cancel_q_ = nullptr; } } tab_reset;
    VK_TRY(ctx->h_out_d.ensure(rq.nq * rq.k * 4));
    VK_TRY(ctx->h_out_l.ensure(rq.nq * rq.k * 8));
    VK_TRY(ctx->h_out_n.ensure(rq.nq * 4 + 64));
    const uint32_t *d_cancel = nullptr, *d_cancel_q = nullptr;
    VK_TRY(ctx->arm_cancel(rq.cancel_flag, &d_cancel, rq.member_cancel ? rq.nq : 0, &d_cancel_q));
    cancel_q_ = d_cancel_q;   // (consumed and cleared by launch(), like tab_)
    if (rq.nq * rq.k <= kZeroCopyEntries) {   // the kernel writes the answer into the pinned host buffers
      VK_TRY(launch(ctx, ctx->d_q.as<float>(), rq.nq, rq.k, rq.ef, d_allow, allow_nbits, ctx->h_out_d.as<float>(),
                    ctx->h_out_l.as<uint64_t>(), ctx->h_out_n.as<uint32_t>(), ctx->stream, true, false, d_cancel));
    } else {
      VK_TRY(ctx->d_out_d.ensure(rq.nq * rq.k * 4));
      VK_TRY(ctx->d_out_l.ensure(rq.nq * rq.k * 8));
      VK_TRY(ctx->d_out_n.ensure(rq.nq * 4));
      VK_TRY(launch(ctx, ctx->d_q.as<float>(), rq.nq, rq.k, rq.ef, d_allow, allow_nbits, ctx->d_out_d.as<float>(),
                    ctx->d_out_l.as<uint64_t>(), ctx->d_out_n.as<uint32_t>(), ctx->stream, true, false, d_cancel));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_d.p, ctx->d_out_d.p, rq.nq * rq.k * 4, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_l.p, ctx->d_out_l.p, rq.nq * rq.k * 8, hipMemcpyDeviceToHost, ctx->stream));
      VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_n.p, ctx->d_out_n.p, rq.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    const size_t st_off = (rq.nq * 4 + 7) & ~(size_t)7;
    VK_HIP_TRY(hipMemcpyAsync(ctx->h_out_n.as<char>() + st_off, ctx->d_stats.p, 40, hipMemcpyDeviceToHost, ctx->stream));
    VK_TRY(ctx->wait(rq.cancel_flag, rq.member_cancel, rq.nq));   // (a raised flag stops the kernel between hops: hnswalg.h:400-402)
    {
      const unsigned long long *st = reinterpret_cast<const unsigned long long *>(ctx->h_out_n.as<char>() + st_off);
      last_n_eval_ = st[0];
      last_n_hops_ = st[1];
      total_n_eval_.fetch_add(st[0], std::memory_order_relaxed);
      total_n_hops_.fetch_add(st[1], std::memory_order_relaxed);
      last_overflow_ = st[2];
      last_redo_ = st[4];
      // (cannot happen: the LDS frontier holds at most 2*ef live entries, the graph-sized one every node)
      if (st[2]) return Status::Err(VK_ERR_INTERNAL, "HNSW search: frontier entries were dropped");
    }
    if (cancel_raised(rq.cancel_flag) && !rq.partial_ok)
      return Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
    for (uint64_t q = 0; q < rq.nq; ++q) {
      uint32_t n = ctx->h_out_n.as<uint32_t>()[q];
      out_n[q] = n;
      memcpy(out_dist + q * rq.k, ctx->h_out_d.as<float>() + q * rq.k, (size_t)n * 4);
      memcpy(out_label + q * rq.k, ctx->h_out_l.as<uint64_t>() + q * rq.k, (size_t)n * 8);
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
    // a context per call; the work is still in flight on the caller's stream when the call returns, the context's
    // next user is ordered behind it (SearchCtx::busy)
    CtxLease lease(pool_, stream);
    SearchCtx *ctx = lease.ctx;
    hipStream_t s = stream ? stream : ctx->stream;
    if (pub_.count == 0) {   // an empty shard answers with empty lists
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
    // (a sharded index hands its members' device-resident filters down: one per query, pointed at where they lie on THIS device)
    if (rq.filter_tab) {
      SearchRequest only = rq;
      only.allow_tab = nullptr;   // (host bitmaps cannot come through a device-buffer call)
      Status ft = build_filter_table(ctx, only, s, &tab_, &tab_nbits_);
      if (!ft.ok()) { tab_ = nullptr; tab_nbits_ = nullptr; (void)ctx->end_async(s); return ft; }
    }
    std::call_once(totals_once_, [&] {
      if (d_totals_.ensure(16).ok() && hipMemset(d_totals_.p, 0, 16) == hipSuccess) totals_zeroed_.store(true);
    });
    device_totals_ = totals_zeroed_.load() ? d_totals_.as<unsigned long long>() : nullptr;
    Status st = launch(ctx, dq, rq.nq, rq.k, rq.ef, rq.allow_bits, rq.allow_nbits, d_out_dist, d_out_label,
                       d_out_n, s, true, false, rq.cancel_word);
    device_totals_ = nullptr;
    tab_ = nullptr;
    tab_nbits_ = nullptr;
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
    VK_TRY(ctx->h_idx.ensure(n * 4));
    VK_TRY(ctx->h_tmp.ensure(n * 4));
    uint32_t *idx = ctx->h_idx.as<uint32_t>();
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) {
      uint32_t id;
      // tombstoned labels are "not found" (vector_hnsw.cc:55-64); so are labels not published to the device yet
      found[i] = graph_->lookup(labels[i], &id) && id < pub_.count && !graph_->is_deleted(id);
      if (found[i]) idx[m++] = id;
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
    VK_TRY(search_labels(query, 1, &label, 1, &d, &l, &n));
    if (n != 1) return Status::Err(VK_ERR_NOT_FOUND, "Couldn't find internal id");
    *out = d;
    return Status::Ok();
  }

  Status get_row(uint64_t label, float *out) override {
    for (;;) {
      bool in_bulk = false;
      {
        std::lock_guard<std::mutex> pl(pend_.mu);
        auto it = pend_.pos.find(label);
        if (it != pend_.pos.end()) {
          memcpy(out, pend_.rows.data() + it->second * params_.dim, (size_t)params_.dim * 4);
          return Status::Ok();
        }
        in_bulk = pend_.linking.count(label) != 0;
      }
      if (!in_bulk) break;
      wait_for_bulk();   // (then look again: see remove())
    }
    std::shared_lock<std::shared_mutex> lk(rw_);
    uint32_t id;
    if (!graph_->lookup(label, &id)) return Status::Err(VK_ERR_NOT_FOUND, "label not found");
    memcpy(out, graph_->row(id), (size_t)params_.dim * 4);
    return Status::Ok();
  }

  Status contains(uint64_t label, bool *found) override {
    std::shared_lock<std::shared_mutex> lk(rw_);   // (resize reallocates the tables under the unique lock)
    uint32_t id;
    {
      std::lock_guard<std::mutex> pl(pend_.mu);
      if (pend_.pos.count(label) || pend_.linking.count(label)) { *found = true; return Status::Ok(); }
    }
    *found = graph_->lookup(label, &id) && !graph_->is_deleted(id);
    return Status::Ok();
  }

  Status stats(vk_index_stats *out) override {
    std::shared_lock<std::shared_mutex> lk(rw_);
    memset(out, 0, sizeof(*out));
    size_t staged;
    {
      std::lock_guard<std::mutex> pl(pend_.mu);
      staged = pend_.live;
    }
    out->count = graph_->count() + staged + draining_.load(std::memory_order_relaxed);
    out->deleted = graph_->deleted_count();
    out->staged_adds = staged_adds_.load(std::memory_order_relaxed);
    out->staged_adds_device = staged_adds_device_.load(std::memory_order_relaxed);
    out->capacity = graph_->max_elements();
    out->device_bytes = store_.device_bytes() + d_links0_.cap + d_upper_slot_.cap + d_upper_pool_.cap;
    out->host_bytes = store_.host_bytes() + graph_->host_bytes();
    out->staged_ops = store_.staged_ops() + staged;
    out->max_level = graph_->max_level();
    out->entry_point = graph_->entry_point();
    out->last_n_eval = last_n_eval_;
    out->last_n_hops = last_n_hops_;
    out->last_frontier_redo = last_redo_;
    out->last_frontier_dropped = last_overflow_;
    out->total_n_eval = total_n_eval_.load(std::memory_order_relaxed);
    out->total_n_hops = total_n_hops_.load(std::memory_order_relaxed);
    if (totals_zeroed_.load() && d_totals_.p) {   // + what device-buffer calls (a sharded index's fan-outs) added on the device so far
      unsigned long long t[2] = {0, 0};
      (void)hipSetDevice(store_.device());
      if (hipMemcpy(t, d_totals_.p, 16, hipMemcpyDeviceToHost) == hipSuccess) {
        out->total_n_eval += t[0];
        out->total_n_hops += t[1];
      }
    }
    // what a tombstone keeps alive until the slot is reused: its row and its level-0 list, on the device and on the host
    // (the reference adds the vector's bytes to reclaimable_memory at markDelete, hnswalg.h:1199)
    out->tombstoned_bytes = out->deleted * ((uint64_t)store_.row_bytes() + (uint64_t)(2 * params_.m + 1) * 4);
    out->max_label = graph_->max_label();
    out->last_visited_mode = last_visited_mode_.load(std::memory_order_relaxed);
    return Status::Ok();
  }

  void filter_devices(std::vector<int> *out) const override { out->assign(1, store_.device()); }

  Status device_rows(uint64_t, void **, uint64_t *) override {
    return Status::Err(VK_ERR_INVALID, "device bulk load is a FLAT-only path (the HNSW graph is built from host rows)");
  }
  Status commit_device_rows(uint64_t, const uint64_t *) override {
    return Status::Err(VK_ERR_INVALID, "device bulk load is a FLAT-only path (the HNSW graph is built from host rows)");
  }

  Status save(vk_write_chunk_fn fn, void *user) override;
  Status load_from(vk_read_chunk_fn fn, void *user);

 private:
  Status add_one(uint64_t label, const float *row) {
    uint32_t id = 0;
    std::vector<float> rounded;
    if (store_.bf16()) {   // the host graph must see what the device will see: rows rounded to bf16
      rounded.resize(params_.dim);
      for (uint32_t i = 0; i < params_.dim; ++i) {
        uint32_t u;
        memcpy(&u, row + i, 4);
        if (!((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu))) u += 0x7FFFu + ((u >> 16) & 1u);
        u &= 0xFFFF0000u;
        memcpy(&rounded[i], &u, 4);
      }
      row = rounded.data();
    }
    VK_TRY(graph_->add(row, label, &id));
    std::lock_guard<std::mutex> g(store_mu_);
    return store_.stage_write(id, row, label);
  }

  Status flush_if_dirty() {
    VK_TRY(drain_pending());
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      if (!store_.dirty() && !graph_->any_dirty()) return Status::Ok();
    }
    std::unique_lock<std::shared_mutex> lk(rw_);
    return flush_locked();
  }

  // What the device mirror holds as of the last flush.  Searches run on THIS, not on the live host graph: add() and
  // remove() hold only the shared lock, so they may run between a flush and a kernel launch (or during the kernel) --
  // a new entry point, a higher level or a larger count whose rows / lists were never published must not reach the
  // launch arguments.
  struct Published { uint32_t count = 0, entry_point = HnswGraph::kNone; int max_level = -1; uint64_t deleted = 0; };
  Published pub_;

  Status flush_locked() {
    (void)hipSetDevice(store_.device());
    VK_TRY(store_.flush());
    if (!graph_->any_dirty() && !links_stale_) { publish_snapshot(); return Status::Ok(); }
    // (flush_links consumes the dirty marks before it touches the device: a failure in there -- an allocation under memory
    //  pressure drops the table it wanted to grow -- leaves the device's copy of the links behind the host's with nothing
    //  marked.  Until one pass has succeeded the next one republishes everything from the host graph, which is the truth.)
    const bool everything = links_stale_;
    links_stale_ = true;
    Status st = flush_links(everything);
    if (st.ok()) {
      links_stale_ = false;
      publish_snapshot();
    }
    return st;
  }
  bool links_stale_ = false;   // under rw_ exclusive, like flush_locked itself

  void publish_snapshot() {   // caller holds rw_ exclusively: no add / remove is running
    pub_.count = (uint32_t)graph_->count();
    pub_.entry_point = graph_->entry_point();
    pub_.max_level = graph_->max_level();
    pub_.deleted = graph_->deleted_count();
  }

  Status flush_links(bool everything) {
    graph_->clear_any_dirty();
    const uint32_t count = (uint32_t)graph_->count();
    const uint32_t l0s = (uint32_t)graph_->maxM0() + 1, ups = (uint32_t)graph_->maxM() + 1;
    const uint64_t cap = std::max<uint64_t>(store_.alloc_rows(), count);
    hipStream_t s = store_.stream();
    bool full = everything;
    if (d_links0_.cap < cap * l0s * 4 || d_upper_slot_.cap < cap * 4) {
      VK_TRY(d_links0_.ensure(cap * l0s * 4));       // DevBuf::ensure drops the old contents:
      VK_TRY(d_upper_slot_.ensure(cap * 4));         // republish everything
      full = true;
    }
    const uint64_t slots = graph_->upper_slots_used();
    if (d_upper_pool_.cap < std::max<uint64_t>(slots, 1) * ups * 4) {
      VK_TRY(d_upper_pool_.ensure(std::max<uint64_t>(slots + slots / 2, 64) * ups * 4));
      full = true;
    }
    std::vector<uint32_t> dl0, dup;
    for (uint32_t i : graph_->take_dirty_ids()) {
      if (i >= count) continue;
      uint8_t f = graph_->take_dirty(i);
      if (full) continue;
      if (f & HnswGraph::kDirtyL0) dl0.push_back(i);
      if (f & HnswGraph::kDirtyUpper) dup.push_back(i);
    }
    if (full || dl0.size() > count / 8) {
      VK_HIP_TRY(hipMemcpyAsync(d_links0_.p, graph_->links0(0), (size_t)count * l0s * 4, hipMemcpyHostToDevice, s));
      dl0.clear();
    }
    if (full) {
      dup.clear();
      for (uint32_t i = 0; i < count; ++i)
        if (graph_->level_of(i) > 0) dup.push_back(i);
      std::vector<uint32_t> slots_host(count);
      for (uint32_t i = 0; i < count; ++i) slots_host[i] = graph_->level_of(i) > 0 ? graph_->upper_slot(i) : HnswGraph::kNone;
      VK_HIP_TRY(hipMemcpyAsync(d_upper_slot_.p, slots_host.data(), (size_t)count * 4, hipMemcpyHostToDevice, s));
      VK_HIP_TRY(hipStreamSynchronize(s));
    }
    // staged scatter of the dirty lists
    PinBuf h_pay, h_idx;
    DevBuf d_pay, d_idx;
    auto scatter = [&](uint32_t *dst, uint32_t stride, size_t n, auto fill) -> Status {
      if (!n) return Status::Ok();
      VK_TRY(h_pay.ensure(n * stride * 4));
      VK_TRY(h_idx.ensure(n * 4));
      VK_TRY(d_pay.ensure(n * stride * 4));
      VK_TRY(d_idx.ensure(n * 4));
      fill(h_pay.as<uint32_t>(), h_idx.as<uint32_t>());
      VK_HIP_TRY(hipMemcpyAsync(d_pay.p, h_pay.p, n * stride * 4, hipMemcpyHostToDevice, s));
      VK_HIP_TRY(hipMemcpyAsync(d_idx.p, h_idx.p, n * 4, hipMemcpyHostToDevice, s));
      VK_HIP_TRY(launch_scatter_u32(dst, d_pay.as<uint32_t>(), d_idx.as<uint32_t>(), (uint32_t)n, stride, s));
      VK_HIP_TRY(hipStreamSynchronize(s));
      return Status::Ok();
    };
    Status st = scatter(d_links0_.as<uint32_t>(), l0s, dl0.size(), [&](uint32_t *pay, uint32_t *idx) {
      for (size_t t = 0; t < dl0.size(); ++t) {
        idx[t] = dl0[t];
        memcpy(pay + t * l0s, graph_->links0(dl0[t]), (size_t)l0s * 4);
      }
    });
    if (st.ok()) {
      size_t nrec = 0;
      for (uint32_t id : dup) nrec += (size_t)graph_->level_of(id);
      st = scatter(d_upper_pool_.as<uint32_t>(), ups, nrec, [&](uint32_t *pay, uint32_t *idx) {
        size_t t = 0;
        for (uint32_t id : dup)
          for (int lv = 1; lv <= graph_->level_of(id); ++lv, ++t) {
            idx[t] = graph_->upper_slot(id) + (uint32_t)(lv - 1);
            memcpy(pay + t * ups, graph_->upper(id, lv), (size_t)ups * 4);
          }
      });
    }
    if (st.ok() && !full)
      st = scatter(d_upper_slot_.as<uint32_t>(), 1, dup.size(), [&](uint32_t *pay, uint32_t *idx) {
        for (size_t t = 0; t < dup.size(); ++t) { idx[t] = dup[t]; pay[t] = graph_->upper_slot(dup[t]); }
      });
    h_pay.release();
    h_idx.release();
    d_pay.release();
    d_idx.release();
    VK_HIP_TRY(hipStreamSynchronize(s));
    return st;
  }

  // One filter per query: every distinct HOST bitmap goes to the device once, device-resident filters (filter_set.hpp) are
  // pointed at where they lie; the kernel gets [nq] pointers and lengths.  (The table is staged in a vector: a copy from
  // pageable memory has consumed its source when the call returns, so this also serves search_device, which returns with
  // its work in flight.)
  Status build_filter_table(SearchCtx *ctx, const SearchRequest &rq, hipStream_t s, const uint64_t *const **d_tab_out,
                            const uint64_t **d_nbits_out) {
    *d_tab_out = nullptr;
    *d_nbits_out = nullptr;
    if (!rq.allow_tab && !rq.filter_tab) return Status::Ok();
    std::vector<const uint64_t *> uniq;
    std::vector<uint64_t> uniq_bits, off;
    size_t words = 0;
    std::vector<uint32_t> which(rq.nq, ~0u);
    std::vector<const uint64_t *> resident(rq.nq, nullptr);
    bool any = false;
    for (uint64_t q = 0; q < rq.nq; ++q) {
      if (rq.filter_tab && rq.filter_tab[q]) {
        resident[q] = rq.filter_tab[q]->bits_on(store_.device());
        if (!resident[q]) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's device");
        any = true;
        continue;
      }
      if (!rq.allow_tab || !rq.allow_tab[q]) continue;
      uint32_t u = 0;
      for (; u < uniq.size(); ++u)
        if (uniq[u] == rq.allow_tab[q] && uniq_bits[u] == rq.allow_nbits_tab[q]) break;
      if (u == uniq.size()) {
        uniq.push_back(rq.allow_tab[q]);
        uniq_bits.push_back(rq.allow_nbits_tab[q]);
        off.push_back(words);
        words += (size_t)((rq.allow_nbits_tab[q] + 63) / 64) + 1;
      }
      which[q] = u;
      any = true;
    }
    if (!any) return Status::Ok();   // (every entry was "no filter")
    const size_t tab_bytes = rq.nq * 16;
    VK_TRY(ctx->d_allow_tab.ensure(tab_bytes + words * 8 + 8));
    char *base = ctx->d_allow_tab.as<char>();
    std::vector<uint64_t> h(2 * rq.nq);
    for (uint64_t q = 0; q < rq.nq; ++q) {
      if (resident[q]) {
        h[q] = reinterpret_cast<uint64_t>(resident[q]);
        h[rq.nq + q] = rq.filter_tab[q]->nbits();
      } else {
        h[q] = which[q] == ~0u ? 0 : reinterpret_cast<uint64_t>(base + tab_bytes + off[which[q]] * 8);
        h[rq.nq + q] = which[q] == ~0u ? 0 : rq.allow_nbits_tab[q];
      }
    }
    VK_HIP_TRY(hipMemcpyAsync(base, h.data(), tab_bytes, hipMemcpyHostToDevice, s));
    for (size_t u = 0; u < uniq.size(); ++u) {
      const size_t w = (size_t)((uniq_bits[u] + 63) / 64);
      if (w) VK_HIP_TRY(hipMemcpyAsync(base + tab_bytes + off[u] * 8, uniq[u], w * 8, hipMemcpyHostToDevice, s));
    }
    *d_tab_out = reinterpret_cast<const uint64_t *const *>(base);
    *d_nbits_out = reinterpret_cast<const uint64_t *>(base + rq.nq * 8);
    return Status::Ok();
  }

  Status launch(SearchCtx *ctx, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef_runtime, const uint64_t *d_allow,
                uint64_t allow_nbits, float *d_out_d, uint64_t *d_out_l, uint32_t *d_out_n, hipStream_t s,
                bool reset_stats, bool out_ids = false, const uint32_t *d_cancel = nullptr) {
    uint64_t ef = ef_runtime ? ef_runtime : graph_->ef();
    ef = std::max<uint64_t>(ef, k);                       // hnswalg.h:1705,1710
    const int e = hnsw_slots_per_lane(ef);
    if (e == 0) return Status::Err(VK_ERR_INVALID, "ef (or k) > 16384 is not served by this build of the HNSW search");
    if (graph_->maxM0() > 4096) return Status::Err(VK_ERR_INVALID, "M > 2048 is not served by this build of the HNSW search");
    const uint32_t count = pub_.count;
    HnswSearchArgs a{};
    a.rows = store_.d_rows();
    a.labels = store_.d_labels();
    a.links0 = d_links0_.as<uint32_t>();
    a.upper_slot = d_upper_slot_.as<uint32_t>();
    a.upper_pool = d_upper_pool_.as<uint32_t>();
    a.queries = d_q;
    a.allow_bits = d_allow;
    a.allow_nbits = allow_nbits;
    a.out_dist = d_out_d;
    a.out_label = d_out_l;
    a.out_n = d_out_n;
    a.row_stride_f = a.q_stride_f = store_.stride_f();
    a.chunks = store_.stride_f() / 16;
    a.l0_stride = (uint32_t)graph_->maxM0() + 1;
    a.up_stride = (uint32_t)graph_->maxM() + 1;
    a.entry_point = pub_.entry_point;
    a.max_level = pub_.max_level;
    a.n_nodes = count;
    a.bitmap_words = (((count + 31) / 32) + 3) & ~3u;
    a.nq = (uint32_t)nq;
    a.k = (uint32_t)k;
    a.ef = (uint32_t)ef;
    a.cand_cap = (uint32_t)std::max<uint64_t>(cand_floor_, 2 * ef);   // frontier pool (LDS): cannot outgrow 2*ef without a filter
    // With a filter or tombstones the result list fills slowly and the frontier grows like the reference's unbounded
    // candidate_set (hnswalg.h:367-370; about ef / selectivity entries): it moves to HBM.  First launch: sized by the
    // graph (every node enters it at most once), capped at 64k entries per wave; a query that outgrows the cap is
    // abandoned there and answered by a second launch whose frontier IS graph-sized (fewer waves; see below), so no
    // search is ever truncated.
    a.allow_tab = tab_;
    a.allow_nbits_tab = tab_nbits_;
    tab_ = nullptr;          // (set by search() around its launch() calls only)
    tab_nbits_ = nullptr;
    // (a result list beyond 2048 entries takes the LDS the frontier would need: the frontier moves to HBM then, too)
    //
    // A FEW tombstones (up to 1 / 16 of the nodes) and no filter: the frontier grows by about their share, and what keeps the
    // result list from filling is rare -- such a batch takes the optimistic launch below (frontier in LDS, visited set on chip)
    // like an index without deletions; a query whose LDS frontier does fill up is given up there and answered by the launch
    // with the graph-sized frontier (the redo list every hash-set launch carries), so nothing is ever truncated.  One
    // deleted key in 1.25M used to cost every search of the index 21 % (the HBM-frontier kernel without any filter: 0.58 of
    // the HBM peak against 0.74, profiles/r06_hnsw_gpool_probe.log).  Only where that launch serves the batch at all (below:
    // a batch that fills the device, a graph large enough for the hash sets); small batches keep the HBM frontier.
    const bool filtered = d_allow != nullptr || a.allow_tab != nullptr;
    auto hash_set_log2 = [&]() -> uint32_t {   // the table of the hash-set launch, 0 = that launch does not serve this batch
      if (visited_hash_ == 0 || !(visited_hash_ == 2 || !hnsw_uses_latency_variant(a))) return 0;
      uint32_t lg = 14;      // (a search evaluates about 25 x ef nodes on the graphs measured: 64 x ef words, at least 64 KB)
      while (lg < 17 && ((uint64_t)1 << lg) < hash_per_ef_ * ef) ++lg;
      if (hash_log2_forced_) lg = hash_log2_forced_;
      return visited_hash_ == 2 || ((uint64_t)8 << lg) <= (uint64_t)a.bitmap_words * 4 ? lg : 0;   // (at most half the bitmap's size)
    };
    const bool few_tombstones = pub_.deleted > 0 && !filtered && ef <= 2048 && (uint64_t)pub_.deleted * 16 <= count &&
                                opt_.get(kOptHnswOptimisticTombstones) != 0 && hash_set_log2() != 0;
    const bool gpool = filtered || (pub_.deleted > 0 && !few_tombstones) || ef > 2048;
    a.gpool_level = gpool ? 1 : 0;
    if (gpool)   // (a multiple of 128: the kernel keeps one minimum per 64 entries in the LDS words of the pool)
      a.cand_cap = (uint32_t)std::min<uint64_t>(gpool_cap(), (std::max<uint64_t>(a.cand_cap, count) + 127) & ~(uint64_t)127);
    bool redo = gpool && a.cand_cap < count;
    a.pool_g = gpool ? reinterpret_cast<float *>(8) : nullptr;   // (placeholder until the buffer is sized below)
    a.nbr_cap = (uint32_t)((graph_->maxM0() + 63) & ~(size_t)63);
    a.check_deleted = pub_.deleted ? 1 : 0;
    a.out_ids = out_ids ? 1 : 0;
    a.cancel = d_cancel;
    a.cancel_q = cancel_q_;
    cancel_q_ = nullptr;
    if (hnsw_lds_bytes(a) > 160 * 1024)
      return Status::Err(VK_ERR_INVALID, "query block + result list (dimension, ef, M) do not fit the 160 KiB of LDS");
    int max_blocks = 0;
    VK_HIP_TRY(hnsw_max_blocks(a, l2(), store_.bf16(), e, &max_blocks));
    // visited sets: one per resident wave, bounded to 4 GiB per search context (VK_HNSW_VISITED_BYTES).  A bitmap of the graph -- or, for a batch that
    // fills the device on a graph large enough for it to matter, an exact hash set of the ids the search touches
    // (typically 25 x ef of them): 64 KB instead of 1.25 MB per wave at 10M nodes, ef = 128, so every wave the CUs can
    // hold gets one (the bitmaps' 2 GiB allowed 1 636 of 4 096), clearing it costs nothing and its atomics hit in cache.
    // A query that outgrows its table is abandoned there and answered by a second launch with bitmaps.
    const uint64_t wpb = (uint64_t)hnsw_waves_per_block(a);
    uint64_t blocks = std::min<uint64_t>((nq + wpb - 1) / wpb, (uint64_t)max_blocks);
    const uint64_t bm_bytes = (uint64_t)a.bitmap_words * 4;
    const uint32_t hash_log2 = gpool ? 0 : hash_set_log2();
    HnswSearchArgs h{};
    uint64_t blocks_h = 0, wpb_h = wpb;
    if (hash_log2) {
      h = a;
      h.vis_hash_log2 = hash_log2;
      h.vis_mode = (uint32_t)visited_mode_;
      h.bitmap_words = 1u << hash_log2;
      // (its LDS frontier need not hold 2 x ef entries for the worst run of equal distances: a query that fills it is
      // re-run like one that fills its table, and the LDS saved is resident waves at large ef)
      h.cand_cap = (uint32_t)((std::max<uint64_t>(cand_floor_, ef + std::max<uint64_t>(64, ef / 4)) + 3) & ~(uint64_t)3);
      if (h.vis_mode >= 3) {
        // the set in LDS (12 KB per wave, ~5500 ids before most new ones spill into the table in memory): ids below 2^24, result
        // lists in registers (up to sixteen slots per lane), two blocks of four waves must fit a CU (at 768 dimensions up to
        // ef = 544, with the frontier trimmed, below), and -- mode 3, the default -- ef x maxM0 within the option
        // hnsw-lds-visited-work: a search evaluates about 0.82 ef maxM0 nodes (10M x 768, M = 16: +15..27 % at ef = 128,
        // +12..17 % at ef = 320 .. 448 where half of the ids spill, +10 % at 480 .. 544: profiles/r05_hnsw_large_ef_lds5*.log).
        // Beyond that eight waves no longer fit (six were -7 % at 640 .. 768).  Mode 4 takes an LDS set whenever it FITS (tests).
        const bool forced = h.vis_mode == 4;
        const uint64_t work = ef * (uint64_t)graph_->maxM0();
        h.vis_mode = 3;                                     // the 12 KB set: two blocks of four waves per CU
        bool fits_small = count < (1u << 24) && e <= 16 && 2 * hnsw_lds_bytes(h) <= 160 * 1024;
        if (!fits_small && count < (1u << 24) && e <= 16) {
          // (ef = 512 misses the budget by 270 B per wave: a frontier of ef + 64 entries instead of ef + ef / 4 -- it holds
          //  candidates nearer than the ef-th result, rarely more than ef of them; a query that fills it is re-run -- and
          //  with ef + 32 entries ef = 544 fits to the byte)
          const uint32_t was = h.cand_cap;
          for (uint64_t room : {(uint64_t)64, (uint64_t)32}) {
            h.cand_cap = (uint32_t)((std::max<uint64_t>(cand_floor_, ef + room) + 3) & ~(uint64_t)3);
            fits_small = 2 * hnsw_lds_bytes(h) <= 160 * 1024;
            if (fits_small) break;
          }
          if (!fits_small) h.cand_cap = was;
        }
        const uint32_t cap_small = h.cand_cap;
        h.vis_mode = 5;                                     // the 32 KB set: one block of four waves per CU
        const bool fits_big = count < (1u << 24) && e <= 8 && hnsw_waves_per_block(h) == 4 && hnsw_lds_bytes(h) <= 160 * 1024;
        if (fits_small && (forced || work <= opt_.get(kOptHnswLdsWork))) h.vis_mode = 3;
        else if (fits_big && (forced || work <= opt_.get(kOptHnswLdsWorkBig))) h.vis_mode = 5;
        else h.vis_mode = 0;
        if (h.vis_mode != 3 && cap_small != 0)   // (the trimmed frontier is the LDS set's concession only)
          h.cand_cap = (uint32_t)((std::max<uint64_t>(cand_floor_, ef + std::max<uint64_t>(64, ef / 4)) + 3) & ~(uint64_t)3);
        // (the table in memory stays: ids that find no room on chip spill into it)
      }
      int mbh = 0;
      VK_HIP_TRY(hnsw_max_blocks(h, l2(), store_.bf16(), e, &mbh));
      // (the hash kernels' own waves per block: with the option hnsw-visited-hash = 2 a batch of up to 1024 queries comes
      //  here too, and `wpb` above is then the latency kernel's ONE wave per block -- sized with it, a batch of 769 .. 1024
      //  queries got 768 tables for up to 1024 waves: a memory fault in the device build of a small graph, found in r04)
      wpb_h = (uint64_t)hnsw_waves_per_block(h);
      blocks_h = std::min<uint64_t>((nq + wpb_h - 1) / wpb_h, (uint64_t)mbh);
      blocks_h = std::max<uint64_t>(1, std::min<uint64_t>(blocks_h, visited_bytes_ / ((uint64_t)h.bitmap_words * 4 * wpb_h)));
      VK_TRY(ctx->d_redo.ensure((nq + 1) * 4));
    }
    blocks = std::max<uint64_t>(1, std::min<uint64_t>(blocks, visited_bytes_ / (bm_bytes * wpb)));
    if (gpool) {   // ... and the HBM frontiers to 4 GiB (VK_HNSW_POOL_BYTES; 1 GiB held half the waves a CU can run: 29.9k -> 36.0k QPS on the hybrid shard)
      blocks = std::max<uint64_t>(1, std::min<uint64_t>(blocks, pool_bytes_ / ((uint64_t)a.cand_cap * 8 * wpb)));
      VK_TRY(ctx->d_pool.ensure(blocks * wpb * (uint64_t)a.cand_cap * 8));
      a.pool_g = ctx->d_pool.as<float>();
    }
    // the second launch (graph-sized frontier: cand_cap >= count, a multiple of 8192; per wave the entries plus one
    // minimum per 64 of them): as many waves as its memory budget holds
    HnswSearchArgs b{};
    uint64_t blocks2 = 0, wpb2 = 0;
    if (redo || hash_log2) {
      b = a;
      b.gpool_level = 2;
      b.cand_cap = (uint32_t)(((uint64_t)count + 8191) & ~(uint64_t)8191);
      if (hnsw_lds_bytes(b) > 160 * 1024) return Status::Err(VK_ERR_INVALID, "graph too large for the LDS frontier index");
      int mb2 = 0;
      VK_HIP_TRY(hnsw_max_blocks(b, l2(), store_.bf16(), e, &mb2));
      wpb2 = (uint64_t)hnsw_waves_per_block(b);
      const uint64_t per_wave = (uint64_t)b.cand_cap * 8 + (uint64_t)b.cand_cap / 64 * 4;
      blocks2 = std::min<uint64_t>((nq + wpb2 - 1) / wpb2, (uint64_t)mb2);
      blocks2 = std::max<uint64_t>(1, std::min<uint64_t>(blocks2, redo_bytes_ / (per_wave * wpb2)));
      blocks2 = std::max<uint64_t>(1, std::min<uint64_t>(blocks2, visited_bytes_ / (bm_bytes * wpb2)));
      VK_TRY(ctx->d_pool2.ensure(blocks2 * wpb2 * per_wave));
      VK_TRY(ctx->d_redo.ensure((nq + 1) * 4));
      b.pool_g = ctx->d_pool2.as<float>();
    }
    // visited-set scratch of the launches that actually run: hash tables (h) + the re-run's bitmaps (b) on the hash
    // path -- the bitmap launch `a` is skipped there, and its blocks * wpb bitmaps are up to 4 GiB per context on
    // exactly the large-graph, large-batch case the hash sets exist for -- else a (+ b)
    const uint64_t vis_first = hash_log2 ? blocks_h * wpb_h * (uint64_t)h.bitmap_words * 4 : blocks * wpb * bm_bytes;
    VK_TRY(ctx->d_tmp.ensure(std::max(vis_first, blocks2 * wpb2 * bm_bytes)));
    a.visited = ctx->d_tmp.as<uint32_t>();
    VK_TRY(ctx->d_stats.ensure(64));
    if (reset_stats) VK_HIP_TRY(hipMemsetAsync(ctx->d_stats.p, 0, 40, s));
    a.stats = ctx->d_stats.as<unsigned long long>();
    a.totals = device_totals_;   // (set by search_device only: the host entry adds what it reads back)
    device_totals_ = nullptr;
    a.queue = reinterpret_cast<uint32_t *>(a.stats + 6);          // [0]: first launch, [1]: second
    VK_HIP_TRY(hipMemsetAsync(a.queue, 0, 8, s));
    if (redo) {
      a.redo_out = ctx->d_redo.as<uint32_t>();
      VK_HIP_TRY(hipMemsetAsync(a.redo_out, 0, 4, s));
    }
    if (hash_log2) {
      // first the launch with the hash sets; then, over the queries it gave up, the kernel that nothing can overflow:
      // bitmaps and a frontier sized by the graph (a few microseconds when there are none)
      h.visited = a.visited;
      h.stats = a.stats;
      h.queue = a.queue;
      h.redo_out = ctx->d_redo.as<uint32_t>();
      VK_HIP_TRY(hipMemsetAsync(h.redo_out, 0, 4, s));
      last_visited_mode_.store(h.vis_mode >= 3 ? h.vis_mode : (h.vis_mode == 2 ? 2 : 1), std::memory_order_relaxed);
      VK_HIP_TRY(launch_hnsw_search(h, l2(), store_.bf16(), e, (uint32_t)blocks_h, s));
      b.visited = a.visited;
      b.stats = a.stats;
      b.queue = a.queue + 1;
      b.redo_out = nullptr;
      b.redo_in = ctx->d_redo.as<uint32_t>();
      VK_HIP_TRY(launch_hnsw_search(b, l2(), store_.bf16(), e, (uint32_t)blocks2, s));
      return Status::Ok();
    }
    last_visited_mode_.store(0, std::memory_order_relaxed);
    VK_HIP_TRY(launch_hnsw_search(a, l2(), store_.bf16(), e, (uint32_t)blocks, s));
    if (redo) {   // a few microseconds when the list is empty
      b.visited = a.visited;
      b.stats = a.stats;
      b.queue = a.queue + 1;
      b.redo_out = nullptr;
      b.redo_in = ctx->d_redo.as<uint32_t>();
      VK_HIP_TRY(launch_hnsw_search(b, l2(), store_.bf16(), e, (uint32_t)blocks2, s));
    }
    return Status::Ok();
  }

  // ---- K9: device-assisted bulk insert (hnsw_build.hip) ---------------------------------------------
  // Level 0 of a batch of NEW points is linked on the device: efConstruction beam search
  // (hnsw_search_kernel, ids out), neighbour selection, reverse links with re-pruning; the host
  // registers the points, links their upper levels (1/M of them) and keeps its level-0 table in
  // step by reading back the lists the batch touched.  *handled = false: preconditions not met,
  // nothing was changed, the caller takes the host path.
  static constexpr uint64_t kDeviceBuildMinBatch = 4096;
  static constexpr uint64_t kDeviceBuildMinGraph = 16384;  // the first points are inserted by the host builder
  static constexpr uint64_t kDeviceBuildMaxBatch = 8192;

  Status add_batch_device(const uint64_t *labels_in, const float *rows, uint64_t n, bool *handled) {
    *handled = false;
    if (graph_->ef_construction() > 512 || graph_->maxM0() > 192) return Status::Ok();
    if ((size_t)store_.stride_f() * 4 * 4 + 8192 > 160 * 1024) return Status::Ok();
    std::vector<uint64_t> iota;
    const uint64_t *labels = labels_in;
    if (!labels) {
      iota.resize(n);
      for (uint64_t i = 0; i < n; ++i) iota[i] = i;
      labels = iota.data();
    }
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      if (!graph_->bulk_possible(labels, n)) return Status::Ok();
    }
    *handled = true;
    uint64_t pos = 0;
    if (graph_->count() < build_min_graph_) {
      pos = std::min<uint64_t>(n, build_min_graph_ - graph_->count());
      VK_TRY(add_batch_host(labels, rows, pos));
    }
    std::unique_lock<std::shared_mutex> lk(rw_);
    (void)hipSetDevice(store_.device());
    bt_ = BuildTimes{};
    // batch b+1 is registered and staged on the host while the device links batch b
    auto batch_size = [&](uint64_t at) {
      return std::min<uint64_t>(n - at, std::max<uint64_t>(build_min_batch_, std::min<uint64_t>(graph_->count() / build_frac_, build_max_batch_)));
    };
    uint32_t first = 0;
    uint64_t P = pos < n ? batch_size(pos) : 0;
    if (P) VK_TRY(prepare_batch(labels + pos, rows + pos * params_.dim, (uint32_t)P, &first));
    while (P) {
      const uint64_t npos = pos + P;
      uint32_t nfirst = 0;
      uint64_t nP = 0;
      auto prepare_next = [&]() -> Status {
        if (npos >= n) return Status::Ok();
        nP = batch_size(npos);
        return prepare_batch(labels + npos, rows + npos * params_.dim, (uint32_t)nP, &nfirst);
      };
      VK_TRY(device_batch(first, (uint32_t)P, prepare_next));
      pos = npos;
      P = nP;
      first = nfirst;
    }
    if (opt_.get(kOptHnswBuildVerbose))
      fprintf(stderr,
              "[vk] device build: %llu batches; register+flush %.2fs, beam search %.2fs, select+group+relink+gather %.2fs, "
              "host table update %.2fs, waiting for the host's upper-level linking %.2fs; beam-search pool overflows %llu, "
              "evals/point %.0f\n",
              (unsigned long long)bt_.batches, bt_.reg, bt_.search, bt_.select, bt_.relink, bt_.upper,
              (unsigned long long)bt_.overflow, bt_.points ? (double)bt_.evals / bt_.points : 0.0);
    return Status::Ok();
  }

  struct BuildTimes { double reg = 0, search = 0, select = 0, relink = 0, upper = 0; uint64_t batches = 0, overflow = 0, evals = 0, points = 0; };
  BuildTimes bt_;
  static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
  }

  // host half of a batch: slots, labels, levels, rows into the graph and the row store's staging area
  Status prepare_batch(const uint64_t *labels, const float *rows, uint32_t P, uint32_t *first_out) {
    const uint32_t dim = params_.dim;
    uint32_t first = 0;
    const double t0 = now_s();
    std::vector<float> rounded;
    if (store_.bf16()) {   // the host graph must hold what the device holds: rows rounded to bf16 (as add_one does)
      rounded.resize((size_t)P * dim);
      for (size_t i = 0; i < rounded.size(); ++i) {
        uint32_t u;
        memcpy(&u, rows + i, 4);
        if (!((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu))) u += 0x7FFFu + ((u >> 16) & 1u);
        u &= 0xFFFF0000u;
        memcpy(&rounded[i], &u, 4);
      }
      rows = rounded.data();
    }
    VK_TRY(graph_->bulk_register(rows, labels, P, &first));
    for (uint32_t i = 0; i < P; ++i) VK_TRY(store_.stage_write(first + i, rows + (size_t)i * dim, labels[i]));
    *first_out = first;
    bt_.reg += now_s() - t0;
    return Status::Ok();
  }

  // device half: the batch [first, first + P) is registered and staged; `prepare_next` (the host half of the
  // following batch) runs while the device works
  template <class PrepareNext>
  Status device_batch(uint32_t first, uint32_t P, PrepareNext &&prepare_next) {
    const uint32_t M = (uint32_t)graph_->M(), maxM0 = (uint32_t)graph_->maxM0();
    const uint32_t l0s = maxM0 + 1;
    double t0 = now_s();
    bt_.batches += 1;
    bt_.points += P;
    VK_TRY(flush_locked());          // rows, labels, the new (empty) lists, upper lists of earlier batches
    // (test failpoint: what an allocation failure of the scratch below looks like -- the batch is registered, nothing of it linked)
    if (const uint64_t f = opt_.get(kOptHnswFailpointBatch); f != 0 && bt_.batches == f)
      return Status::Err(VK_ERR_INTERNAL, "failpoint: device batch");
    CtxLease lease(pool_);
    SearchCtx *ctx = lease.ctx;
    hipStream_t s = ctx->stream;
    bt_.reg += now_s() - t0;
    t0 = now_s();
    // A: candidates = efConstruction nearest live nodes of every new point, ascending
    const uint32_t efc = (uint32_t)graph_->ef_construction();
    VK_TRY(ctx->d_out_d.ensure((size_t)P * efc * 4));
    VK_TRY(ctx->d_out_l.ensure((size_t)P * efc * 8));
    VK_TRY(ctx->d_out_n.ensure((size_t)P * 4));
    const float *d_new = static_cast<const float *>(store_.d_rows()) + (size_t)first * store_.stride_f();
    if (store_.bf16()) {   // the beam search takes f32 queries: widen the new rows into a query block
      VK_TRY(ctx->d_q.ensure((size_t)P * store_.stride_f() * 4));
      VK_HIP_TRY(launch_hnsw_widen_rows(store_.d_rows(), store_.stride_f(), first, P, ctx->d_q.as<float>(), s));
      d_new = ctx->d_q.as<float>();
    }
    VK_TRY(launch(ctx, d_new, P, efc, efc, nullptr, 0, ctx->d_out_d.as<float>(), ctx->d_out_l.as<uint64_t>(),
                  ctx->d_out_n.as<uint32_t>(), s, true, true));
    if (opt_.get(kOptHnswBuildVerbose)) {
      unsigned long long st[4];
      VK_HIP_TRY(hipMemcpyAsync(st, ctx->d_stats.p, 32, hipMemcpyDeviceToHost, s));
      VK_HIP_TRY(hipStreamSynchronize(s));
      bt_.evals += st[0];
      bt_.overflow += st[2];
      bt_.search += now_s() - t0;
      t0 = now_s();
    }
    // B..E on the device without a host round trip: neighbour selection (writes the new points' own
    // lists), grouping of the (new point, neighbour) pairs by neighbour (radix sort), reverse links,
    // and a compact copy of every list the batch wrote for the host's table
    const size_t np = (size_t)P * M;                       // pairs, upper bound
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t tmp_bytes = hnsw_group_tmp_bytes((uint32_t)np);
    const size_t lists_words = ((size_t)P + np) * l0s;
    size_t o_sel_id = 0, o_sel_d = o_sel_id + al(np * 4), o_sel_n = o_sel_d + al(np * 4), o_keys_a = o_sel_n + al((size_t)P * 4),
           o_keys_b = o_keys_a + al(np * 8), o_vals_a = o_keys_b + al(np * 8), o_add_p = o_vals_a + al(np * 4),
           o_add_d = o_add_p + al(np * 4), o_flags = o_add_d + al(np * 4), o_pos = o_flags + al(np * 4),
           o_off = o_pos + al(np * 4), o_tmp = o_off + al((np + 1) * 4), o_ret = o_tmp + al(tmp_bytes);
    // returned to the host in one copy: [counts 2 | node np | lists]
    const size_t ret_bytes = al(8) + al(np * 4) + lists_words * 4;
    VK_TRY(ctx->d_part_d.ensure(o_ret + ret_bytes));
    char *db = ctx->d_part_d.as<char>();
    uint32_t *d_counts = reinterpret_cast<uint32_t *>(db + o_ret);
    uint32_t *d_node = reinterpret_cast<uint32_t *>(db + o_ret + al(8));
    uint32_t *d_lists = reinterpret_cast<uint32_t *>(db + o_ret + al(8) + al(np * 4));
    HnswBuildArgs b{};
    b.rows = store_.d_rows();
    b.row_stride_f = store_.stride_f();
    b.chunks = store_.stride_f() / 16;
    b.links0 = d_links0_.as<uint32_t>();
    b.l0_stride = l0s;
    b.max_keep = M;
    b.cand_id = ctx->d_out_l.as<uint64_t>();
    b.cand_dist = ctx->d_out_d.as<float>();
    b.cand_n = ctx->d_out_n.as<uint32_t>();
    b.cand_ld = efc;
    b.n_new = P;
    b.first_id = first;
    b.sel_id = reinterpret_cast<uint32_t *>(db + o_sel_id);
    b.sel_dist = reinterpret_cast<float *>(db + o_sel_d);
    b.sel_n = reinterpret_cast<uint32_t *>(db + o_sel_n);
    VK_HIP_TRY(launch_hnsw_select(b, l2(), store_.bf16(), s));
    HnswGroupArgs g{};
    g.sel_id = b.sel_id;
    g.sel_dist = b.sel_dist;
    g.sel_n = b.sel_n;
    g.n_new = P;
    g.m = M;
    g.first_id = first;
    g.keys_a = reinterpret_cast<uint64_t *>(db + o_keys_a);
    g.keys_b = reinterpret_cast<uint64_t *>(db + o_keys_b);
    g.vals_a = reinterpret_cast<uint32_t *>(db + o_vals_a);
    g.add_p = reinterpret_cast<uint32_t *>(db + o_add_p);
    g.add_d = reinterpret_cast<float *>(db + o_add_d);
    g.flags = reinterpret_cast<uint32_t *>(db + o_flags);
    g.pos = reinterpret_cast<uint32_t *>(db + o_pos);
    g.off = reinterpret_cast<uint32_t *>(db + o_off);
    g.node = d_node;
    g.counts = d_counts;
    g.tmp = db + o_tmp;
    g.tmp_bytes = tmp_bytes;
    VK_HIP_TRY(launch_hnsw_group(g, s));
    b.node = g.node;
    b.off = g.off;
    b.add_p = g.add_p;
    b.add_d = g.add_d;
    b.counts = g.counts;
    b.n_touched = (uint32_t)np;
    b.max_keep = maxM0;
    VK_HIP_TRY(launch_hnsw_relink(b, l2(), store_.bf16(), s));
    VK_HIP_TRY(launch_hnsw_gather_lists(d_lists, d_links0_.as<uint32_t>(), l0s, first, P, d_node, d_counts, s));
    VK_TRY(ctx->h_q.ensure(ret_bytes));
    VK_HIP_TRY(hipMemcpyAsync(ctx->h_q.p, db + o_ret, ret_bytes, hipMemcpyDeviceToHost, s));
    // while the device works on level 0: the upper levels of the batch's points on the host, in
    // parallel like concurrent addPoint calls (the kernels above were enqueued with the entry point
    // and level of the graph as flushed, so a new entry point made here is seen from the next batch on)
    std::vector<uint32_t> ups;
    for (uint32_t i = 0; i < P; ++i)
      if (graph_->level_of(first + i) > 0) ups.push_back(first + i);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    Status err;
    std::mutex err_mu;
    auto work = [&]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= ups.size() || failed.load()) return;
        Status st = graph_->bulk_link_upper(ups[i]);
        if (!st.ok()) {
          std::lock_guard<std::mutex> g2(err_mu);
          if (!failed.exchange(true)) err = st;
        }
      }
    };
    std::vector<std::thread> pool;
    {
      const unsigned threads = (unsigned)std::min<size_t>(ups.size(), params_.build_threads ? params_.build_threads : effective_cpus());
      for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work);
    }
    struct Joiner {
      std::vector<std::thread> &p;
      ~Joiner() { for (auto &t : p) if (t.joinable()) t.join(); }
    } joiner{pool};
    VK_TRY(prepare_next());          // host work for the next batch, behind the device
    VK_HIP_TRY(hipStreamSynchronize(s));
    bt_.select += now_s() - t0;
    t0 = now_s();
    {   // keep the host table in step with what the batch wrote
      const char *hb = ctx->h_q.as<char>();
      const uint32_t T = reinterpret_cast<const uint32_t *>(hb)[0];
      const uint32_t *node = reinterpret_cast<const uint32_t *>(hb + al(8));
      const uint32_t *src = reinterpret_cast<const uint32_t *>(hb + al(8) + al(np * 4));
      uint32_t *tab = graph_->links0_table();
      const size_t count_now = graph_->count();
      if (T > np) return Status::Err(VK_ERR_INTERNAL, "device build: inconsistent group count");
      for (size_t i = 0; i < (size_t)P + T; ++i) {
        const uint32_t id = i < P ? first + (uint32_t)i : node[i - P];
        if (id >= count_now) return Status::Err(VK_ERR_INTERNAL, "device build: node id out of range");
        memcpy(tab + (size_t)id * l0s, src + i * l0s, (size_t)l0s * 4);
      }
    }
    bt_.relink += now_s() - t0;
    t0 = now_s();
    for (auto &t : pool) t.join();
    if (failed.load()) return err;
    bt_.upper += now_s() - t0;
    return Status::Ok();
  }

  // batch size of the device build: min(count / build_frac_, build_max_batch_) -- the points of one batch
  // do not see one another, so a batch stays small relative to the graph it extends.  Measured
  // (100k x 128, recall@10 at ef=64): 1/8 of the graph per batch 0.9016, 1/32 0.9090, host build 0.9086.
  OptRef build_max_batch_{&opt_, kOptHnswBuildBatch};
  OptRef build_min_graph_{&opt_, kOptHnswBuildMinGraph};
  OptRef build_min_batch_{&opt_, kOptHnswBuildMinBatch};
  OptRef build_frac_{&opt_, kOptHnswBuildFrac};
  OptRef cand_floor_{&opt_, kOptHnswPoolFloor};
  // cap of the first launch's HBM frontier (entries per wave, a multiple of 128) and the memory the second launch's
  // graph-sized frontiers may take per context; hnsw-gpool-cap = 128 forces the second launch on small test graphs
  uint64_t gpool_cap() const { return std::max<uint64_t>(128, opt_.get(kOptHnswGpoolCap) & ~(uint64_t)127); }
  // visited sets as hash tables: 0 never, 1 when it pays (default), 2 always (tests); table words per unit of ef; fixed size
  OptRef visited_hash_{&opt_, kOptHnswVisitedHash};
  OptRef hash_per_ef_{&opt_, kOptHnswHashPerEf};
  OptRef visited_mode_{&opt_, kOptHnswVisitedMode};
  OptRef hash_log2_forced_{&opt_, kOptHnswHashLog2};
  OptRef pool_bytes_{&opt_, kOptHnswPoolBytes};
  OptRef visited_bytes_{&opt_, kOptHnswVisitedBytes};
  OptRef redo_bytes_{&opt_, kOptHnswRedoBytes};
  OptRef device_build_{&opt_, kOptHnswDeviceBuild};
  // new labels staged by single add() calls, linked in bulk by drain_pending()
  struct Pending {
    std::mutex mu;
    std::vector<float> rows;                       // [labels.size()][dim]
    std::vector<uint64_t> labels;
    std::unordered_map<uint64_t, size_t> pos;      // live staged labels -> their place
    std::unordered_map<uint64_t, size_t> linking;  // the labels of the bulk drain_pending() is linking right now
    size_t live = 0;
  } pend_;
  std::atomic<uint64_t> draining_{0}, staged_adds_{0}, staged_adds_device_{0}, last_visited_mode_{0};
  std::mutex drain_mu_;
  std::vector<float> spare_rows_;
  std::vector<uint64_t> spare_labels_;
  bool stage_candidate() const {
    return opt_.get(kOptHnswStageAdds) != 0 && device_build_ != 0 && graph_->count() >= build_min_graph_ &&
           !(params_.allow_replace_deleted && graph_->deleted_count() > 0) && graph_->ef_construction() <= 512 && graph_->maxM0() <= 192;
  }
  RowStore store_;
  CtxPool pool_;
  std::unique_ptr<HnswGraph> graph_;
  std::shared_mutex rw_;
  std::mutex store_mu_;
  DevBuf d_links0_, d_upper_slot_, d_upper_pool_;
  std::atomic<uint64_t> last_n_eval_{0}, last_n_hops_{0}, last_overflow_{0}, last_redo_{0}, total_n_eval_{0}, total_n_hops_{0};
  static thread_local const uint64_t *const *tab_;
  static thread_local const uint64_t *tab_nbits_;
  static thread_local const uint32_t *cancel_q_;
  static thread_local unsigned long long *device_totals_;
  DevBuf d_totals_;
  std::atomic<bool> totals_zeroed_{false};
  std::once_flag totals_once_;
};

thread_local const uint64_t *const *HnswIndex::tab_ = nullptr;
thread_local const uint64_t *HnswIndex::tab_nbits_ = nullptr;
thread_local const uint32_t *HnswIndex::cancel_q_ = nullptr;
thread_local unsigned long long *HnswIndex::device_totals_ = nullptr;

// ---- persistence: hnswalg.h:808-865 (SaveIndex), :887-1139 (LoadIndex + loadCheck) -----------------
Status HnswIndex::save(vk_write_chunk_fn fn, void *user) {
  VK_TRY(drain_pending());   // (staged rows are part of the index)
  std::shared_lock<std::shared_mutex> lk(rw_);
  const HnswGraph &g = *graph_;
  const size_t vec = (size_t)params_.dim * 4;
  const size_t sl0 = g.maxM0() * 4 + 4, slu = g.maxM() * 4 + 4;
  const size_t off_data = (sl0 + 7) & ~(size_t)7;
  std::string hdr;
  pb_put_varint_field(hdr, 2, g.max_elements());
  pb_put_varint_field(hdr, 3, g.count());
  pb_put_varint_field(hdr, 4, sl0 + vec + 8);
  pb_put_varint_field(hdr, 5, off_data + 8);
  pb_put_varint_field(hdr, 6, sl0);
  pb_put_varint_field(hdr, 7, (uint64_t)(int64_t)g.max_level());   // int32: negatives are 10-byte varints
  pb_put_varint_field(hdr, 8, g.count() ? g.entry_point() : 0xFFFFFFFFull);
  pb_put_varint_field(hdr, 9, g.maxM());
  pb_put_varint_field(hdr, 10, g.maxM0());
  pb_put_varint_field(hdr, 11, g.M());
  pb_put_double_field(hdr, 12, g.mult());
  pb_put_varint_field(hdr, 13, g.ef_construction());
  if (fn(user, hdr.data(), hdr.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
  if (g.count() == 0) return Status::Ok();
  std::vector<char> buf(sl0 + vec + 8);
  for (uint32_t i = 0; i < g.count(); ++i) {
    memcpy(buf.data(), g.links0(i), sl0);
    memcpy(buf.data() + sl0, g.row(i), vec);
    uint64_t lab = g.label_of(i);
    memcpy(buf.data() + sl0 + vec, &lab, 8);
    if (fn(user, buf.data(), buf.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
  }
  for (uint32_t i = 0; i < g.count(); ++i) {
    uint64_t sz = g.level_of(i) > 0 ? slu * (size_t)g.level_of(i) : 0;
    if (fn(user, &sz, 8)) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
    if (sz && fn(user, g.upper(i, 1), sz)) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
  }
  return Status::Ok();
}

#define VK_LOAD_CHECK(ok, msg) \
  if (!(ok)) return Status::Err(VK_ERR_INTERNAL, std::string("HNSW index load validation failed: ") + (msg))
// ... a graph-invariant check whose violation the device kernels survive: skipped under the kill switch
// (vk_index_params.load_skip_validation = the reference's hnsw-validation-enable no, hnswalg.h:872-885)
#define VK_LOAD_CHECK_SOFT(ok, msg) \
  if (!skip_soft) VK_LOAD_CHECK(ok, msg)

Status HnswIndex::load_from(vk_read_chunk_fn fn, void *user) {
  const size_t vec = (size_t)params_.dim * 4;
  const bool skip_soft = params_.load_skip_validation != 0;
  std::vector<char> buf(std::max<size_t>(vec + 8 + 4 + 8 * 10000, 4096));
  uint64_t len = 0;
  if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
  // (label_offset and offset_data, fields 5 and 6, are not read at all: like the reference -- hnswalg.h:921-930 -- the
  // geometry is recomputed from M and the vector size, so a snapshot written before the 8-byte padding of the record
  // (offset_data = 132 at M = 16, testing/vector_test.cc:764-800) loads like a new one)
  uint64_t f[14] = {0};
  double mult = 0;
  int64_t max_level = 0;
  {
    PbReader r{reinterpret_cast<const uint8_t *>(buf.data()), reinterpret_cast<const uint8_t *>(buf.data()) + len};
    uint32_t field, wire;
    uint64_t val;
    while (r.next(&field, &wire, &val)) {
      if (field == 12 && wire == 1) memcpy(&mult, &val, 8);
      else if (field == 7) max_level = (int32_t)(uint32_t)val;
      else if (field < 14) f[field] = val;
    }
  }
  const size_t M = f[11], maxM = f[9], maxM0 = f[10], cur = f[3];
  const size_t exp_m = params_.m > 10000 ? 10000 : params_.m;
  VK_LOAD_CHECK(exp_m >= 1, "M must be >= 1");
  VK_LOAD_CHECK(M == exp_m, "header M does not match index definition");
  VK_LOAD_CHECK(maxM == M, "header maxM does not equal M");
  VK_LOAD_CHECK(maxM0 == 2 * M, "header maxM0 does not equal 2*M");
  VK_LOAD_CHECK(maxM0 <= 0xFFFF, "maxM0 exceeds the 16-bit neighbor-count field");
  const size_t sl0 = maxM0 * 4 + 4, slu = maxM * 4 + 4;
  VK_LOAD_CHECK(f[4] == sl0 + vec + 8, "serialized element size is inconsistent with the geometry");
  VK_LOAD_CHECK(f[1] == 0, "offset_level_0 must be 0");
  if (M >= 2) {
    const double expected = 1.0 / log((double)M);
    VK_LOAD_CHECK_SOFT(mult > 0.0 && fabs(mult - expected) <= 1e-6 * expected, "mult is inconsistent with M");
  }
  const size_t max_elements = std::max<size_t>(cur, std::max<size_t>(params_.initial_cap, f[2]));
  VK_LOAD_CHECK(cur <= max_elements, "curr_element_count exceeds max_elements");
  VK_LOAD_CHECK(max_elements < (1ull << 32), "max_elements out of range");
  const uint32_t enterpoint = (uint32_t)f[8];
  if (cur == 0) {
    VK_LOAD_CHECK(max_level == -1 || max_level == 0, "empty index has a non-trivial max_level");
  } else {
    VK_LOAD_CHECK(max_level >= 0, "non-empty index has a negative max_level");
    VK_LOAD_CHECK(max_level <= (int64_t)cur, "max_level exceeds the element count");
    VK_LOAD_CHECK(enterpoint < cur, "enterpoint_node is out of range");
  }
  std::unique_lock<std::shared_mutex> lk(rw_);
  graph_ = std::make_unique<HnswGraph>(params_.dim, params_.metric == VK_METRIC_L2, max_elements, M,
                                       f[13] ? f[13] : params_.ef_construction, params_.random_seed,
                                       params_.allow_replace_deleted != 0);
  graph_->set_ef(params_.ef_runtime ? params_.ef_runtime : 10);   // ef_runtime is not persisted (vector_hnsw.cc:159-160)
  if (buf.size() < sl0 + vec + 8 + slu * 64) buf.resize(sl0 + vec + 8 + slu * 64);
  for (uint32_t i = 0; i < cur; ++i) {
    if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
    VK_LOAD_CHECK(len == sl0 + vec + 8, "level-0 element chunk has the wrong size");
    const uint32_t *ll = reinterpret_cast<const uint32_t *>(buf.data());
    const size_t cnt = ll[0] & 0xFFFFu;
    VK_LOAD_CHECK(cnt <= maxM0, "level-0 neighbor count exceeds 2*M");
    for (size_t jn = 0; jn < cnt; ++jn) {
      VK_LOAD_CHECK(ll[1 + jn] < cur, "level-0 neighbor id out of range");
      VK_LOAD_CHECK_SOFT(ll[1 + jn] != i, "level-0 self-loop");
    }
    uint64_t lab;
    memcpy(&lab, buf.data() + sl0 + vec, 8);
    VK_TRY(observe_loaded_row(lab, buf.data() + sl0));
    VK_TRY(graph_->load_element(i, ll, reinterpret_cast<const float *>(buf.data() + sl0), lab));
    VK_TRY(store_.stage_write(i, reinterpret_cast<const float *>(buf.data() + sl0), lab));
    if (store_.staged_bytes() >= ((size_t)256 << 20)) VK_TRY(store_.flush());
  }
  for (uint32_t i = 0; i < cur; ++i) {
    if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
    VK_LOAD_CHECK(len == 8, "link-list size chunk has the wrong size");
    uint64_t sz;
    memcpy(&sz, buf.data(), 8);
    if (sz == 0) continue;
    VK_LOAD_CHECK(sz % slu == 0, "upper-level link-list size is not a multiple of the stride");
    const int64_t level = (int64_t)(sz / slu);
    VK_LOAD_CHECK(level <= max_level, "element level exceeds max_level");
    if (buf.size() < sz) buf.resize(sz);
    if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
    VK_LOAD_CHECK(len == sz, "upper-level link-list chunk has the wrong size");
    const uint32_t *w = reinterpret_cast<const uint32_t *>(buf.data());
    for (int64_t l = 0; l < level; ++l) VK_LOAD_CHECK((w[l * (maxM + 1)] & 0xFFFFu) <= maxM, "upper-level neighbor count exceeds M");
    VK_TRY(graph_->load_upper(i, w, sz / 4));
  }
  // (under the kill switch a shorter entry point is still refused when it would send the descent through levels it
  // does not have; a TALLER graph than the header says only means levels the search never enters)
  if (cur > 0) {
    VK_LOAD_CHECK_SOFT(graph_->level_of(enterpoint) == max_level, "enterpoint node is not at max_level");
    VK_LOAD_CHECK(graph_->level_of(enterpoint) >= max_level, "enterpoint node is not at max_level");
  }
  for (uint32_t i = 0; i < cur; ++i)
    for (int level = 1; level <= graph_->level_of(i); ++level) {
      const uint32_t *ll = graph_->upper(i, level);
      const size_t cnt = ll[0] & 0xFFFFu;
      for (size_t jn = 0; jn < cnt; ++jn) {
        const uint32_t e = ll[1 + jn];
        VK_LOAD_CHECK(e < cur, "upper-level neighbor id out of range");
        VK_LOAD_CHECK_SOFT(e != i, "upper-level self-loop");
        VK_LOAD_CHECK(graph_->level_of(e) >= level, "upper-level neighbor is absent at that level");
      }
    }
  Status dup = graph_->load_labels(cur);
  if (!dup.ok()) return dup;
  graph_->load_finish(cur, cur ? (int)max_level : -1, cur ? enterpoint : HnswGraph::kNone);
  return flush_locked();
}

static Status pick_device_h(const vk_index_params &p, int *device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return Status::Err(VK_ERR_NO_DEVICE, "no HIP device: libvkindex needs a gfx950 GPU (no CPU fallback)");
  int d = p.device_id;
  if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
  if (d >= n) return Status::Err(VK_ERR_INVALID, "device_id out of range");
  *device = d;
  return Status::Ok();
}

Status create_hnsw(const vk_index_params &p, std::unique_ptr<Index> *out) {
  int device = 0;
  VK_TRY(pick_device_h(p, &device));
  *out = std::make_unique<HnswIndex>(p, device);
  return Status::Ok();
}

Status load_hnsw(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out) {
  int device = 0;
  VK_TRY(pick_device_h(p, &device));
  auto ix = std::make_unique<HnswIndex>(p, device);
  VK_TRY(ix->load_from(fn, user));
  *out = std::move(ix);
  return Status::Ok();
}

}  // namespace vk
