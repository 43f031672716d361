// vk_index.cc -- the C ABI declared in include/vk_index.h: argument checking, error
// strings, and dispatch to FlatIndex / HnswIndex.  No exception leaves this file.
#include "../../include/vk_index.h"

#include <string.h>

#include <exception>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

#include <atomic>
#include <chrono>

#include "dispatcher.hpp"
#include "index.hpp"

// cumulative counters behind vk_index_stats (INFO fields and metrics of the adaptor: vector_base.cc:385-409,
// metrics.h:40-50,75-80, latency samples search.cc:149,160)
struct VkCounters {
  std::atomic<uint64_t> searches{0}, calls{0}, lat_sum_ns{0};
  std::atomic<uint64_t> errors[VK_STATUS_COUNT];
  std::atomic<uint64_t> hist[16];
  VkCounters() {
    for (auto &e : errors) e.store(0);
    for (auto &h : hist) h.store(0);
  }
  // one search call answering nq queries after ns nanoseconds with status code
  void record(uint64_t nq, int code, uint64_t ns) {
    calls.fetch_add(1, std::memory_order_relaxed);
    if (code != VK_OK) {
      errors[code < VK_STATUS_COUNT ? code : VK_ERR_INTERNAL].fetch_add(1, std::memory_order_relaxed);
      return;
    }
    searches.fetch_add(nq, std::memory_order_relaxed);
    lat_sum_ns.fetch_add(ns * nq, std::memory_order_relaxed);
    const uint64_t us = ns / 1000;
    int b = 0;
    while (b < 15 && us >= ((uint64_t)64 << b)) ++b;
    hist[b].fetch_add(nq, std::memory_order_relaxed);
  }
};

// a device-resident filter behind the ABI: a counted reference on a vk::FilterSet (filter_set.hpp)
struct vk_filter {
  std::shared_ptr<vk::FilterSet> set;
  const vk_index *owner;
  std::atomic<uint32_t> refs{1};
};

// filters under caller keys (vk_index_filter_cache_*): least recently used first out, bounded in entries and device bytes
struct VkFilterCache {
  struct Entry { std::string key; uint64_t epoch; std::shared_ptr<vk::FilterSet> set; };
  std::mutex mu;
  std::list<Entry> lru;   // front = most recently used
  std::unordered_map<std::string, std::list<Entry>::iterator> by_key;
  uint64_t bytes = 0;
  std::atomic<uint64_t> hits{0}, misses{0}, built{0};
  void drop(std::list<Entry>::iterator it) {
    bytes -= it->set->device_bytes();
    by_key.erase(it->key);
    lru.erase(it);
  }
};

struct vk_index {
  std::unique_ptr<vk::Index> impl;
  VkCounters counters;
  VkFilterCache filters;
  std::atomic<vk_batch_done_fn> batch_done{nullptr};   // vk_index_set_batch_completion
  void *batch_user = nullptr;
  std::unique_ptr<vk::Dispatcher> dispatcher;   // (declared after impl: destroyed first, while the index is still there)
};

thread_local const vk::LoadObserver *vk::g_load_observer = nullptr;

namespace {
thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
int done(const vk::Status &s) {
  if (s.ok()) return VK_OK;
  g_last_error = s.msg;
  return s.code;
}

template <class F>
int guarded(F &&f) {
  try {
    return done(f());
  } catch (const std::exception &e) {
    return fail(VK_ERR_INTERNAL, e.what());
  } catch (...) {
    return fail(VK_ERR_INTERNAL, "unknown exception");
  }
}

// a search entry point: status + the cumulative counters
template <class F>
int counted(vk_index *ix, uint64_t nq, F &&f) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = guarded(f);
  const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  ix->counters.record(nq, rc, ns);
  return rc;
}

// the serving options live in the dispatcher
void sync_dispatcher(vk_index *ix) {
  const vk::Options &o = ix->impl->options();
  ix->dispatcher->configure((uint32_t)o.get(vk::kOptCoalesceMaxBatch), (uint32_t)o.get(vk::kOptCoalesceMaxWaitUs));
  ix->dispatcher->set_in_flight((uint32_t)o.get(vk::kOptBatchesInFlight));
  ix->dispatcher->set_queue_depth(o.get(vk::kOptMaxQueryQueueDepth));
  ix->dispatcher->set_completers((uint32_t)o.get(vk::kOptCompleterThreads), (uint32_t)o.get(vk::kOptHandoutChunk));
}

vk::Status check_params(const vk_index_params *p) {
  if (!p) return vk::Status::Err(VK_ERR_INVALID, "params is NULL");
  if (p->struct_size != sizeof(vk_index_params)) return vk::Status::Err(VK_ERR_INVALID, "vk_index_params.struct_size mismatch");
  if (p->algo > VK_ALGO_HNSW) return vk::Status::Err(VK_ERR_INVALID, "unknown algo");
  if (p->metric > VK_METRIC_COSINE) return vk::Status::Err(VK_ERR_INVALID, "unknown metric");
  if (p->dtype > VK_DTYPE_BF16) return vk::Status::Err(VK_ERR_INVALID, "unknown dtype");
  if (p->dim == 0 || p->dim > 64000) return vk::Status::Err(VK_ERR_INVALID, "dimension out of range");
  if (p->initial_cap >= (1ull << 32)) return vk::Status::Err(VK_ERR_INVALID, "initial_cap out of range");
  if (p->algo == VK_ALGO_HNSW && (p->m < 2 || p->m > 10000)) return vk::Status::Err(VK_ERR_INVALID, "M out of range");
  if (p->n_shards > VK_MAX_SHARDS) return vk::Status::Err(VK_ERR_INVALID, "n_shards out of range");
  if (p->shard_ef_pct > 1000) return vk::Status::Err(VK_ERR_INVALID, "shard_ef_pct out of range (0 = 100, at most 1000)");
  if (p->load_skip_validation > 1) return vk::Status::Err(VK_ERR_INVALID, "load_skip_validation must be 0 or 1");
  return vk::Status::Ok();
}
}  // namespace

extern "C" {

const char *vk_last_error(void) { return g_last_error.c_str(); }

uint64_t vk_abi_struct_size(int which) { return which == 0 ? sizeof(vk_index_params) : which == 1 ? sizeof(vk_index_stats) : 0; }

int vk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vk_index_create(const vk_index_params *params, vk_index **out) {
  if (!out) return fail(VK_ERR_INVALID, "out is NULL");
  *out = nullptr;
  return guarded([&]() -> vk::Status {
    VK_TRY(check_params(params));
    std::unique_ptr<vk::Index> impl;
    if (params->n_shards >= 1) VK_TRY(vk::create_sharded(*params, &impl));
    else if (params->algo == VK_ALGO_FLAT) VK_TRY(vk::create_flat(*params, &impl));
    else VK_TRY(vk::create_hnsw(*params, &impl));
    *out = new vk_index;
    (*out)->impl = std::move(impl);
    (*out)->dispatcher = std::make_unique<vk::Dispatcher>((*out)->impl.get());
    sync_dispatcher(*out);
    return vk::Status::Ok();
  });
}

void vk_index_destroy(vk_index *ix) {
  try {
    delete ix;
  } catch (...) {
  }
}

#define VK_NEED(ix)                                             \
  if (!(ix) || !(ix)->impl) return fail(VK_ERR_INVALID, "index is NULL")

// UINT64_MAX is the padding label of every result list ((+inf, UINT64_MAX) past the count, vk_index.h): a row under it would
// be taken for padding by the selection kernels.  The reference's labels are VectorBase's counter from zero.
static const char kReservedLabel[] = "label UINT64_MAX is reserved (the padding of result lists)";
static bool has_reserved_label(const uint64_t *labels, uint64_t n) {
  if (!labels) return false;
  for (uint64_t i = 0; i < n; ++i)
    if (labels[i] == ~0ull) return true;
  return false;
}

int vk_index_add(vk_index *ix, uint64_t label, const void *row) {
  VK_NEED(ix);
  if (!row) return fail(VK_ERR_INVALID, "row is NULL");
  if (label == ~0ull) return fail(VK_ERR_INVALID, kReservedLabel);
  return guarded([&] { return ix->impl->add(label, static_cast<const float *>(row)); });
}

int vk_index_add_batch(vk_index *ix, const uint64_t *labels, const void *rows, uint64_t n) {
  VK_NEED(ix);
  if (n && !rows) return fail(VK_ERR_INVALID, "rows is NULL");
  if (has_reserved_label(labels, n)) return fail(VK_ERR_INVALID, kReservedLabel);   // (before anything of the batch is in)
  return guarded([&] { return ix->impl->add_batch(labels, static_cast<const float *>(rows), n); });
}

int vk_index_remove(vk_index *ix, uint64_t label) {
  VK_NEED(ix);
  return guarded([&] { return ix->impl->remove(label); });
}

int vk_index_resize(vk_index *ix, uint64_t new_max_elements) {
  VK_NEED(ix);
  if (new_max_elements >= (1ull << 32)) return fail(VK_ERR_INVALID, "new_max_elements out of range");
  return guarded([&] { return ix->impl->resize(new_max_elements); });
}

int vk_index_set_ef(vk_index *ix, uint32_t ef) {
  VK_NEED(ix);
  return guarded([&] { return ix->impl->set_ef(ef); });
}

int vk_index_flush(vk_index *ix) {
  VK_NEED(ix);
  return guarded([&] { return ix->impl->flush(); });
}

int vk_index_search_batch(vk_index *ix, const void *queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                          const uint64_t *allow_bits, uint64_t allow_nbits, const volatile int *cancel_flag,
                          int partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
  VK_NEED(ix);
  if (nq && (!queries || !out_n)) return fail(VK_ERR_INVALID, "queries/out_n is NULL");
  if (nq && k && (!out_dist || !out_label)) return fail(VK_ERR_INVALID, "output buffers are NULL");
  return counted(ix, nq, [&] {
    vk::SearchRequest rq;
    rq.queries = static_cast<const float *>(queries);
    rq.nq = nq;
    rq.k = k;
    rq.ef = ef_runtime;
    rq.allow_bits = allow_bits;
    rq.allow_nbits = allow_nbits;
    rq.cancel_flag = cancel_flag;
    rq.partial_ok = partial_ok != 0;
    return ix->impl->search(rq, out_dist, out_label, out_n);
  });
}

int vk_index_search_batch_filters(vk_index *ix, const void *queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                                  const uint64_t *const *allow_bits_tab, const uint64_t *allow_nbits_tab,
                                  const volatile int *cancel_flag, int partial_ok, float *out_dist, uint64_t *out_label,
                                  uint64_t *out_n) {
  VK_NEED(ix);
  if (nq && (!queries || !out_n)) return fail(VK_ERR_INVALID, "queries/out_n is NULL");
  if (nq && k && (!out_dist || !out_label)) return fail(VK_ERR_INVALID, "output buffers are NULL");
  if (nq && allow_bits_tab && !allow_nbits_tab) return fail(VK_ERR_INVALID, "allow_nbits_tab is NULL");
  return counted(ix, nq, [&] {
    vk::SearchRequest rq;
    rq.queries = static_cast<const float *>(queries);
    rq.nq = nq;
    rq.k = k;
    rq.ef = ef_runtime;
    rq.allow_tab = allow_bits_tab;
    rq.allow_nbits_tab = allow_nbits_tab;
    rq.cancel_flag = cancel_flag;
    rq.partial_ok = partial_ok != 0;
    return ix->impl->search(rq, out_dist, out_label, out_n);
  });
}

// A FLAT index serves one scan per distinct allow-bitmap.  The dispatcher groups filtered FLAT requests into lanes by filter
// (one scan per lane: dispatcher.hpp); a BLOCKING call with a raw host bitmap -- which nobody else can share -- is not
// worth a lane of its own and runs directly on one of the index's search contexts, as before.
static bool flat_filtered(vk_index *ix, const uint64_t *allow_bits) {
  return allow_bits && ix->impl->params().algo == VK_ALGO_FLAT;
}

int vk_index_search(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, const uint64_t *allow_bits,
                    uint64_t allow_nbits, const volatile int *cancel_flag, int partial_ok, float *out_dist,
                    uint64_t *out_label, uint64_t *out_n) {
  if (ix && ix->impl && ix->dispatcher->enabled() && k && !flat_filtered(ix, allow_bits) && !vk::cancel_raised(cancel_flag)) {
    if (!query || !out_n || !out_dist || !out_label) return fail(VK_ERR_INVALID, "NULL argument");
    return counted(ix, 1, [&] {
      return ix->dispatcher->search(static_cast<const float *>(query), k, ef_runtime, allow_bits, allow_nbits, nullptr, cancel_flag,
                                    partial_ok != 0, out_dist, out_label, out_n);
    });
  }
  return vk_index_search_batch(ix, query, 1, k, ef_runtime, allow_bits, allow_nbits, cancel_flag, partial_ok,
                               out_dist, out_label, out_n);
}

namespace {
// completion of a submitted request: the caller's callback behind the counters
struct SubmitCtx {
  vk_index *ix;
  vk_search_done_fn done;
  void *user;
  std::chrono::steady_clock::time_point t0;
};
void submit_done(void *p, int status) {
  SubmitCtx *c = static_cast<SubmitCtx *>(p);
  const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c->t0).count();
  c->ix->counters.record(1, status, ns);
  vk_search_done_fn done = c->done;
  void *user = c->user;
  delete c;
  done(user, status);
}
// ... and the same for a span of requests (Dispatcher::set_batch_done): one clock reading, one call of the caller's hook
void submit_done_batch(void *p, void *const *users, const int *statuses, uint64_t n) {
  vk_index *ix = static_cast<vk_index *>(p);
  vk_batch_done_fn hook = ix->batch_done.load(std::memory_order_acquire);
  const auto now = std::chrono::steady_clock::now();
  thread_local std::vector<vk_completion> items;
  items.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    SubmitCtx *c = static_cast<SubmitCtx *>(users[i]);
    ix->counters.record(1, statuses[i], (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - c->t0).count());
    items[i].user = c->user;
    items[i].status = statuses[i];
    items[i].reserved = 0;
    if (!hook) c->done(c->user, statuses[i]);   // (the hook was taken away while requests were in flight)
    delete c;
  }
  if (hook) hook(ix->batch_user, items.data(), n);
}
}  // namespace

int vk_index_set_batch_completion(vk_index *ix, vk_batch_done_fn hook, void *hook_user) {
  VK_NEED(ix);
  ix->batch_user = hook_user;
  ix->batch_done.store(hook, std::memory_order_release);
  ix->dispatcher->set_batch_done(hook ? &submit_done_batch : nullptr, ix);
  return VK_OK;
}

int vk_index_search_submit(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, const uint64_t *allow_bits,
                           uint64_t allow_nbits, const volatile int *cancel_flag, int partial_ok, float *out_dist,
                           uint64_t *out_label, uint64_t *out_n, vk_search_done_fn done, void *user) {
  VK_NEED(ix);
  if (!query || !out_n || !out_dist || !out_label || !done) return fail(VK_ERR_INVALID, "NULL argument");
  if (k == 0) return fail(VK_ERR_INVALID, "k must be positive");
  if (!ix->dispatcher->enabled()) return fail(VK_ERR_INVALID, "vk_index_search_submit needs coalescing (vk_index_set_coalescing with max_batch > 1)");
  return guarded([&]() -> vk::Status {
    SubmitCtx *c = new SubmitCtx{ix, done, user, std::chrono::steady_clock::now()};
    vk::Status st = ix->dispatcher->submit(static_cast<const float *>(query), k, ef_runtime, allow_bits, allow_nbits, nullptr, cancel_flag,
                                           partial_ok != 0, out_dist, out_label, out_n, submit_done, c);
    if (!st.ok()) delete c;   // (not queued: the callback will not fire)
    return st;
  });
}

int vk_index_search_batch_device(vk_index *ix, const void *d_queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                                 const uint64_t *d_allow_bits, uint64_t allow_nbits, float *d_out_dist,
                                 uint64_t *d_out_label, uint32_t *d_out_n, void *hip_stream) {
  VK_NEED(ix);
  if (nq && (!d_queries || !d_out_dist || !d_out_label || !d_out_n)) return fail(VK_ERR_INVALID, "device buffers are NULL");
  return guarded([&] {   // (device-buffer calls return with the work in flight: no latency to record; counted as calls)
    ix->counters.calls.fetch_add(1, std::memory_order_relaxed);
    ix->counters.searches.fetch_add(nq, std::memory_order_relaxed);
    vk::SearchRequest rq;
    rq.queries = static_cast<const float *>(d_queries);
    rq.nq = nq;
    rq.k = k;
    rq.ef = ef_runtime;
    rq.allow_bits = d_allow_bits;
    rq.allow_nbits = allow_nbits;
    return ix->impl->search_device(rq, d_out_dist, d_out_label, d_out_n, static_cast<hipStream_t>(hip_stream));
  });
}

int vk_index_search_labels(vk_index *ix, const void *query, uint64_t k, const uint64_t *labels, uint64_t n_labels,
                           float *out_dist, uint64_t *out_label, uint64_t *out_n) {
  VK_NEED(ix);
  if (!query || !out_n || (n_labels && !labels)) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] {
    return ix->impl->search_labels(static_cast<const float *>(query), k, labels, n_labels, out_dist, out_label, out_n);
  });
}

int vk_index_distance(vk_index *ix, uint64_t label, const void *query, float *out) {
  VK_NEED(ix);
  if (!query || !out) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] { return ix->impl->distance(label, static_cast<const float *>(query), out); });
}

int vk_index_get_row(vk_index *ix, uint64_t label, void *out_row) {
  VK_NEED(ix);
  if (!out_row) return fail(VK_ERR_INVALID, "out_row is NULL");
  return guarded([&] { return ix->impl->get_row(label, static_cast<float *>(out_row)); });
}

int vk_index_contains(vk_index *ix, uint64_t label, int *out_found) {
  VK_NEED(ix);
  if (!out_found) return fail(VK_ERR_INVALID, "out_found is NULL");
  return guarded([&] {
    bool f = false;
    vk::Status s = ix->impl->contains(label, &f);
    *out_found = f ? 1 : 0;
    return s;
  });
}

int vk_index_get_stats(vk_index *ix, vk_index_stats *out) {
  VK_NEED(ix);
  if (!out) return fail(VK_ERR_INVALID, "out is NULL");
  return guarded([&] {
    vk::Status s = ix->impl->stats(out);
    const vk::Dispatcher &d = *ix->dispatcher;
    out->coalesced_batches = d.batches();
    out->coalesced_queries = d.queries();
    out->submitted = d.submitted();
    out->rejected = d.rejected();
    out->queued_now = d.queued();
    out->max_batches_in_flight = d.max_in_flight_seen();
    out->cancelled_early = d.left_early();
    {
      const vk::Dispatcher::Times t = d.times();
      out->dispatch_idle_us = t.idle_us;
      out->dispatch_window_us = t.window_us;
      out->dispatch_search_us = t.search_us;
      out->dispatch_handout_us = t.handout_us;
      out->dispatch_completer_us = t.completer_us;
    }
    {
      VkFilterCache &fc = ix->filters;
      out->filters_built = fc.built.load(std::memory_order_relaxed);
      out->filter_cache_hits = fc.hits.load(std::memory_order_relaxed);
      out->filter_cache_misses = fc.misses.load(std::memory_order_relaxed);
      std::lock_guard<std::mutex> lk(fc.mu);
      out->filter_cache_entries = fc.lru.size();
      out->filter_cache_bytes = fc.bytes;
    }
    const VkCounters &c = ix->counters;
    out->searches = c.searches.load(std::memory_order_relaxed);
    out->search_calls = c.calls.load(std::memory_order_relaxed);
    for (int i = 0; i < VK_STATUS_COUNT; ++i) out->search_errors[i] = c.errors[i].load(std::memory_order_relaxed);
    for (int i = 0; i < 16; ++i) out->latency_hist[i] = c.hist[i].load(std::memory_order_relaxed);
    out->latency_sum_ns = c.lat_sum_ns.load(std::memory_order_relaxed);
    return s;
  });
}

int vk_index_shard_stats(vk_index *ix, uint32_t shard, vk_index_stats *out) {
  VK_NEED(ix);
  if (!out) return fail(VK_ERR_INVALID, "out is NULL");
  return guarded([&] { return ix->impl->shard_stats(shard, out); });
}

int vk_index_set_coalescing(vk_index *ix, uint32_t max_batch, uint32_t max_wait_us) {
  VK_NEED(ix);
  if (max_batch > 16384) return fail(VK_ERR_INVALID, "max_batch out of range");
  return guarded([&]() -> vk::Status {
    VK_TRY(ix->impl->set_option("coalesce-max-batch", max_batch));
    VK_TRY(ix->impl->set_option("coalesce-max-wait-us", max_wait_us));
    sync_dispatcher(ix);
    return vk::Status::Ok();
  });
}

int vk_index_set_option(vk_index *ix, const char *name, uint64_t value) {
  VK_NEED(ix);
  if (!name) return fail(VK_ERR_INVALID, "name is NULL");
  return guarded([&]() -> vk::Status {
    VK_TRY(ix->impl->set_option(name, value));
    sync_dispatcher(ix);
    return vk::Status::Ok();
  });
}

int vk_index_get_option(vk_index *ix, const char *name, uint64_t *out_value) {
  VK_NEED(ix);
  if (!name || !out_value) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] { return ix->impl->get_option(name, out_value); });
}

int vk_index_device_rows(vk_index *ix, uint64_t n_rows, void **d_rows, uint64_t *row_stride_bytes) {
  VK_NEED(ix);
  if (!d_rows || !row_stride_bytes) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] { return ix->impl->device_rows(n_rows, d_rows, row_stride_bytes); });
}

int vk_index_commit_device_rows(vk_index *ix, uint64_t n_rows, const uint64_t *labels) {
  VK_NEED(ix);
  if (has_reserved_label(labels, n_rows)) return fail(VK_ERR_INVALID, kReservedLabel);
  return guarded([&] { return ix->impl->commit_device_rows(n_rows, labels); });
}

int vk_index_shard_count(vk_index *ix, uint32_t *out_n) {
  VK_NEED(ix);
  if (!out_n) return fail(VK_ERR_INVALID, "out_n is NULL");
  *out_n = ix->impl->shard_count();
  return VK_OK;
}

int vk_index_shard_device_rows(vk_index *ix, uint32_t shard, uint64_t n_rows, void **d_rows, uint64_t *row_stride_bytes) {
  VK_NEED(ix);
  if (!d_rows || !row_stride_bytes) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] { return ix->impl->shard_device_rows(shard, n_rows, d_rows, row_stride_bytes); });
}

int vk_index_shard_commit_device_rows(vk_index *ix, uint32_t shard, uint64_t n_rows, const uint64_t *labels) {
  VK_NEED(ix);
  if (has_reserved_label(labels, n_rows)) return fail(VK_ERR_INVALID, kReservedLabel);
  return guarded([&] { return ix->impl->shard_commit_device_rows(shard, n_rows, labels); });
}

int vk_merge_topk_device(const float *d_dist, const uint64_t *d_label, uint32_t parts, uint64_t nq, uint64_t k,
                         float *d_out_dist, uint64_t *d_out_label, uint32_t *d_out_n, int device_id,
                         void *hip_stream) {
  if (!d_dist || !d_label || !d_out_dist || !d_out_label || !d_out_n) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&]() -> vk::Status {
    int e = vk::flat_scan_slots_per_lane(k);
    if (e == 0 || k == 0) return vk::Status::Err(VK_ERR_INVALID, "k out of range for the device merge");
    if (device_id >= 0) VK_HIP_TRY(hipSetDevice(device_id));
    vk::MergeArgs m{};
    m.in_dist = d_dist;
    m.in_label = d_label;
    m.part_stride = nq * k;
    m.q_stride = k;
    m.parts = parts;
    m.per_part = (uint32_t)k;
    m.k = (uint32_t)k;
    m.out_dist = d_out_dist;
    m.out_label = d_out_label;
    m.out_n = d_out_n;
    VK_HIP_TRY(vk::launch_merge_topk(m, e, nq, static_cast<hipStream_t>(hip_stream)));
    return vk::Status::Ok();
  });
}

int vk_index_save(vk_index *ix, vk_write_chunk_fn write_chunk, void *user) {
  VK_NEED(ix);
  if (!write_chunk) return fail(VK_ERR_INVALID, "write_chunk is NULL");
  return guarded([&] { return ix->impl->save(write_chunk, user); });
}

int vk_index_load_tracked(const vk_index_params *params, vk_read_chunk_fn read_chunk, void *user, vk_row_fn on_row, void *row_user,
                          vk_index **out) {
  if (!out || !read_chunk) return fail(VK_ERR_INVALID, "NULL argument");
  *out = nullptr;
  const vk::LoadObserver obs{on_row, row_user};
  struct Scope {   // (the loaders run on this thread)
    explicit Scope(const vk::LoadObserver *o) { vk::g_load_observer = o; }
    ~Scope() { vk::g_load_observer = nullptr; }
  } scope(on_row ? &obs : nullptr);
  return guarded([&]() -> vk::Status {
    VK_TRY(check_params(params));
    std::unique_ptr<vk::Index> impl;
    if (params->n_shards >= 1) VK_TRY(vk::load_sharded(*params, read_chunk, user, &impl));
    else if (params->algo == VK_ALGO_FLAT) VK_TRY(vk::load_flat(*params, read_chunk, user, &impl));
    else VK_TRY(vk::load_hnsw(*params, read_chunk, user, &impl));
    *out = new vk_index;
    (*out)->impl = std::move(impl);
    (*out)->dispatcher = std::make_unique<vk::Dispatcher>((*out)->impl.get());
    sync_dispatcher(*out);
    return vk::Status::Ok();
  });
}

int vk_index_load(const vk_index_params *params, vk_read_chunk_fn read_chunk, void *user, vk_index **out) {
  return vk_index_load_tracked(params, read_chunk, user, nullptr, nullptr, out);
}

// ---- device-resident filters ------------------------------------------------------------------------------------------
int vk_filter_create(vk_index *ix, uint64_t nbits, const uint64_t *labels, uint64_t n_labels, const uint64_t *runs, uint64_t n_runs,
                     const uint64_t *base_bits, vk_filter **out) {
  VK_NEED(ix);
  if (!out) return fail(VK_ERR_INVALID, "out is NULL");
  *out = nullptr;
  return guarded([&]() -> vk::Status {
    std::vector<int> devs;
    ix->impl->filter_devices(&devs);
    std::shared_ptr<vk::FilterSet> set;
    VK_TRY(vk::FilterSet::build(devs, nbits, labels, n_labels, runs, n_runs, base_bits, &set));
    ix->filters.built.fetch_add(1, std::memory_order_relaxed);
    *out = new vk_filter{std::move(set), ix};
    return vk::Status::Ok();
  });
}

int vk_filter_combine(vk_index *ix, const vk_filter *a, const vk_filter *b, uint32_t op, vk_filter **out) {
  VK_NEED(ix);
  if (!a || !b || !out) return fail(VK_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (a->owner != ix || b->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
  return guarded([&]() -> vk::Status {
    std::shared_ptr<vk::FilterSet> set;
    VK_TRY(vk::FilterSet::combine(*a->set, *b->set, op, &set));
    ix->filters.built.fetch_add(1, std::memory_order_relaxed);
    *out = new vk_filter{std::move(set), ix};
    return vk::Status::Ok();
  });
}

int vk_filter_combine_batch(vk_index *ix, const vk_filter *const *a, const vk_filter *const *b, const uint32_t *ops, uint64_t n,
                            vk_filter **out) {
  VK_NEED(ix);
  if (n && (!a || !b || !ops || !out)) return fail(VK_ERR_INVALID, "NULL argument");
  for (uint64_t i = 0; i < n; ++i) {
    out[i] = nullptr;
    if (!a[i] || !b[i]) return fail(VK_ERR_INVALID, "NULL filter");
    if (a[i]->owner != ix || b[i]->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
  }
  return guarded([&]() -> vk::Status {
    std::vector<const vk::FilterSet *> sa(n), sb(n);
    for (uint64_t i = 0; i < n; ++i) { sa[i] = a[i]->set.get(); sb[i] = b[i]->set.get(); }
    std::vector<std::shared_ptr<vk::FilterSet>> sets;
    VK_TRY(vk::FilterSet::combine_batch(sa.data(), sb.data(), ops, n, &sets));
    ix->filters.built.fetch_add(n, std::memory_order_relaxed);
    for (uint64_t i = 0; i < n; ++i) out[i] = new vk_filter{std::move(sets[i]), ix};
    return vk::Status::Ok();
  });
}

void vk_filter_retain(vk_filter *f) {
  if (f) f->refs.fetch_add(1, std::memory_order_relaxed);
}

void vk_filter_release(vk_filter *f) {
  if (!f) return;
  if (f->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    try {
      delete f;   // (the FilterSet itself lives on while a search or the cache holds it)
    } catch (...) {
    }
  }
}

int vk_filter_info(const vk_filter *f, uint64_t *out_nbits, uint64_t *out_allowed) {
  if (!f) return fail(VK_ERR_INVALID, "filter is NULL");
  if (out_nbits) *out_nbits = f->set->nbits();
  if (out_allowed) *out_allowed = f->set->allowed();
  return VK_OK;
}

int vk_filter_read(const vk_filter *f, uint64_t *out_words, uint64_t n_words) {
  if (!f || (n_words && !out_words)) return fail(VK_ERR_INVALID, "NULL argument");
  return guarded([&] { return f->set->read(out_words, n_words); });
}

int vk_index_filter_cache_get(vk_index *ix, const void *key, uint64_t key_len, uint64_t epoch, vk_filter **out) {
  VK_NEED(ix);
  if (!out || (key_len && !key)) return fail(VK_ERR_INVALID, "NULL argument");
  *out = nullptr;
  return guarded([&]() -> vk::Status {
    VkFilterCache &fc = ix->filters;
    std::string k(static_cast<const char *>(key), (size_t)key_len);
    std::lock_guard<std::mutex> lk(fc.mu);
    auto it = fc.by_key.find(k);
    if (it != fc.by_key.end() && it->second->epoch != epoch) {   // stored before the predicate's answer may have changed
      fc.drop(it->second);
      it = fc.by_key.end();
    }
    if (it == fc.by_key.end()) {
      fc.misses.fetch_add(1, std::memory_order_relaxed);
      return vk::Status::Ok();
    }
    fc.lru.splice(fc.lru.begin(), fc.lru, it->second);
    fc.hits.fetch_add(1, std::memory_order_relaxed);
    *out = new vk_filter{it->second->set, ix};
    return vk::Status::Ok();
  });
}

int vk_index_filter_cache_put(vk_index *ix, const void *key, uint64_t key_len, uint64_t epoch, vk_filter *f) {
  VK_NEED(ix);
  if (!f || (key_len && !key)) return fail(VK_ERR_INVALID, "NULL argument");
  if (f->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
  return guarded([&]() -> vk::Status {
    VkFilterCache &fc = ix->filters;
    const uint64_t max_entries = ix->impl->options().get(vk::kOptFilterCacheEntries);
    const uint64_t max_bytes = ix->impl->options().get(vk::kOptFilterCacheBytes);
    std::string k(static_cast<const char *>(key), (size_t)key_len);
    std::lock_guard<std::mutex> lk(fc.mu);
    auto it = fc.by_key.find(k);
    if (it != fc.by_key.end()) fc.drop(it->second);
    if (max_entries == 0) return vk::Status::Ok();   // (the cache is switched off)
    fc.lru.push_front(VkFilterCache::Entry{k, epoch, f->set});
    fc.by_key[k] = fc.lru.begin();
    fc.bytes += f->set->device_bytes();
    while (fc.lru.size() > 1 && (fc.lru.size() > max_entries || fc.bytes > max_bytes)) fc.drop(std::prev(fc.lru.end()));
    return vk::Status::Ok();
  });
}

int vk_index_search_filter(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, vk_filter *filter,
                           const volatile int *cancel_flag, int partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
  VK_NEED(ix);
  if (!filter) return vk_index_search(ix, query, k, ef_runtime, nullptr, 0, cancel_flag, partial_ok, out_dist, out_label, out_n);
  if (filter->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
  if (!query || !out_n || (k && (!out_dist || !out_label))) return fail(VK_ERR_INVALID, "NULL argument");
  if (ix->dispatcher->enabled() && k && !vk::cancel_raised(cancel_flag)) {
    return counted(ix, 1, [&] {
      return ix->dispatcher->search(static_cast<const float *>(query), k, ef_runtime, nullptr, 0, filter->set, cancel_flag, partial_ok != 0,
                                    out_dist, out_label, out_n);
    });
  }
  return counted(ix, 1, [&] {
    vk::SearchRequest rq;
    rq.queries = static_cast<const float *>(query);
    rq.nq = 1;
    rq.k = k;
    rq.ef = ef_runtime;
    rq.filter = filter->set.get();
    rq.cancel_flag = cancel_flag;
    rq.partial_ok = partial_ok != 0;
    return ix->impl->search(rq, out_dist, out_label, out_n);
  });
}

int vk_index_search_submit_filter(vk_index *ix, const void *query, uint64_t k, uint64_t ef_runtime, vk_filter *filter,
                                  const volatile int *cancel_flag, int partial_ok, float *out_dist, uint64_t *out_label, uint64_t *out_n,
                                  vk_search_done_fn done, void *user) {
  VK_NEED(ix);
  if (!query || !out_n || !out_dist || !out_label || !done) return fail(VK_ERR_INVALID, "NULL argument");
  if (k == 0) return fail(VK_ERR_INVALID, "k must be positive");
  if (filter && filter->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
  if (!ix->dispatcher->enabled()) return fail(VK_ERR_INVALID, "vk_index_search_submit needs coalescing (vk_index_set_coalescing with max_batch > 1)");
  return guarded([&]() -> vk::Status {
    SubmitCtx *c = new SubmitCtx{ix, done, user, std::chrono::steady_clock::now()};
    vk::Status st = ix->dispatcher->submit(static_cast<const float *>(query), k, ef_runtime, nullptr, 0, filter ? filter->set : nullptr,
                                           cancel_flag, partial_ok != 0, out_dist, out_label, out_n, submit_done, c);
    if (!st.ok()) delete c;   // (not queued: the callback will not fire)
    return st;
  });
}

int vk_index_search_batch_filter_handles(vk_index *ix, const void *queries, uint64_t nq, uint64_t k, uint64_t ef_runtime,
                                         vk_filter *const *filters, const volatile int *cancel_flag, int partial_ok, float *out_dist,
                                         uint64_t *out_label, uint64_t *out_n) {
  VK_NEED(ix);
  if (nq && (!queries || !out_n)) return fail(VK_ERR_INVALID, "queries/out_n is NULL");
  if (nq && k && (!out_dist || !out_label)) return fail(VK_ERR_INVALID, "output buffers are NULL");
  std::vector<const vk::FilterSet *> tab;
  if (filters) {
    tab.resize(nq);
    for (uint64_t q = 0; q < nq; ++q) {
      if (filters[q] && filters[q]->owner != ix) return fail(VK_ERR_INVALID, "the filter belongs to another index");
      tab[q] = filters[q] ? filters[q]->set.get() : nullptr;
    }
  }
  return counted(ix, nq, [&] {
    vk::SearchRequest rq;
    rq.queries = static_cast<const float *>(queries);
    rq.nq = nq;
    rq.k = k;
    rq.ef = ef_runtime;
    rq.filter_tab = filters ? tab.data() : nullptr;
    rq.cancel_flag = cancel_flag;
    rq.partial_ok = partial_ok != 0;
    return ix->impl->search(rq, out_dist, out_label, out_n);
  });
}

}  // extern "C"
