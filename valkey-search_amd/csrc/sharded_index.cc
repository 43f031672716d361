// sharded_index.cc -- one index over several GPUs of one node, inside the library (one process, as valkey-server is).
//
// The reference scales a vector index by cluster shards: every shard holds an independent index over its slice of the
// keys, a query fans out to all of them and SearchPartitionResultsTracker::AddResult keeps the best k of what comes
// back (src/query/fanout.cc:162-175).  The same shape inside one vk_index: n_shards sub-indexes (FLAT or HNSW, one
// graph per shard), each on its own device -- or several on one device, "logical shards", which is how a one-GPU box
// tests this path.  Rows are dealt to the shards in contiguous runs (a bulk load of N rows puts rows
// [s*N/S, (s+1)*N/S) on shard s); a label stays on the shard it first went to.
//
// A search: the queries (on the serving device = the first shard's) are broadcast to the other devices by peer copy on
// each shard's stream, every shard answers on its own stream (vk_index_search_batch_device of the sub-index: the same
// kernels as a single-device index), the per-shard top-k lists land in one [shard][query][k] array on the serving
// device -- written there directly by a shard that lives on that device, one peer copy per list otherwise (xGMI; 30 KiB
// per shard at B=256, k=10: latency-bound, so one-shot copies rather than a ring collective) -- and merge_topk_kernel
// selects the k best by (distance,label).  That order is total, so the S-shard answer is bit-identical to the
// answer of one index over all rows (FLAT), unlike the reference's arrival-order tie rule (fanout.cc:171).
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>

#include <dlfcn.h>
// RCCL is loaded with dlopen on first use (option shard-gather = 1), so the library builds on a box without the RCCL headers
// too: the handful of types and constants the gather needs are then declared here as the NCCL ABI fixes them (nccl.h:
// ncclResult_t 0 = ncclSuccess; ncclDataType_t: ncclFloat32 = 7, ncclUint64 = 5; ncclComm_t an opaque pointer).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5, ncclFloat32 = 7 } ncclDataType_t;
#endif

#include "index.hpp"
#include "shard_layout.hpp"

// Test seam (link time, not run time): a program that defines this symbol names the library the gather loads instead of the
// system's librccl -- the multi-device pre-flight (tests/helpers/san_sharded_main.cc) carries an in-process model of the five
// entry points.  libvkindex.so does not define it.
extern "C" const char *vk_test_rccl_library() __attribute__((weak));

namespace vk {

namespace {

constexpr uint32_t kMaxShards = 16;
const char kShardMagic[8] = {'V', 'K', 'S', 'H', 'A', 'R', 'D', 'S'};

// per-call resources of a sharded search: per shard a stream, an event and buffers on the shard's device; on the serving
// device the gathered lists; pinned host buffers for the host entry points
struct ShardLane {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  DevBuf d_q, d_allow, d_out_d, d_out_l, d_out_n;
};
// shard-gather = 1: per DISTINCT device the lists of its shards, contiguous ([P][nq][k], P = the most shards any device
// holds; unused slots stay (+inf, no label)), and the all-gathered lists of every device ([G][P][nq][k])
struct GatherGroup {
  int device = 0;
  DevBuf send_d, send_l, recv_d, recv_l;
  size_t filled_nk = 0;       // the pad slots were filled for this nq * k
  hipEvent_t gathered = nullptr;
};
struct MultiCtx {
  std::vector<ShardLane> lane;
  std::vector<GatherGroup> group;
  int dev0 = 0;
  hipStream_t s0 = nullptr;
  hipEvent_t ready = nullptr, busy = nullptr;
  bool has_busy = false;
  DevBuf d_q, d_allow, d_all_d, d_all_l, d_all_n, d_fin_d, d_fin_l, d_fin_n;
  PinBuf h_q, h_fin_d, h_fin_l, h_fin_n, h_cancel;
  ~MultiCtx() {
    (void)hipSetDevice(dev0);
    if (s0) (void)hipStreamSynchronize(s0);
    for (ShardLane &l : lane) {
      (void)hipSetDevice(l.device);
      if (l.stream) (void)hipStreamSynchronize(l.stream);
      for (DevBuf *b : {&l.d_q, &l.d_allow, &l.d_out_d, &l.d_out_l, &l.d_out_n}) b->release();
      if (l.done) (void)hipEventDestroy(l.done);
      if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (GatherGroup &g : group) {
      (void)hipSetDevice(g.device);
      for (DevBuf *b : {&g.send_d, &g.send_l, &g.recv_d, &g.recv_l}) b->release();
      if (g.gathered) (void)hipEventDestroy(g.gathered);
    }
    (void)hipSetDevice(dev0);
    for (DevBuf *b : {&d_q, &d_allow, &d_all_d, &d_all_l, &d_all_n, &d_fin_d, &d_fin_l, &d_fin_n}) b->release();
    for (PinBuf *b : {&h_q, &h_fin_d, &h_fin_l, &h_fin_n, &h_cancel}) b->release();
    if (ready) (void)hipEventDestroy(ready);
    if (busy) (void)hipEventDestroy(busy);
    if (s0) (void)hipStreamDestroy(s0);
  }
};

// One enqueue thread per device.  The reference issues its per-shard requests concurrently (src/query/fanout.cc:69-160);
// here a shard's search is a sequence of a dozen kernel launches (r02: two dozen), and enqueueing shard after shard
// from the calling thread made the host the bottleneck of a fan-out: eight shards x ~40 us before the last device had
// anything to do, against well under a millisecond of device work per shard.  Each worker owns the shards of one device
// and runs the jobs handed to it in order; the caller enqueues the serving device's shards itself and waits on a latch.
class ShardWorkers {
 public:
  struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t left = 0;
    void done() {
      std::lock_guard<std::mutex> g(mu);
      if (--left == 0) cv.notify_one();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return left == 0; });
    }
  };
  explicit ShardWorkers(size_t n) : w_(n) {
    for (size_t i = 0; i < n; ++i) w_[i].th = std::thread([this, i] { run(i); });
  }
  ~ShardWorkers() {
    for (Worker &w : w_) {
      { std::lock_guard<std::mutex> g(w.mu); w.stop = true; }
      w.cv.notify_one();
    }
    for (Worker &w : w_) w.th.join();
  }
  size_t size() const { return w_.size(); }
  void post(size_t i, std::function<void()> job) {
    Worker &w = w_[i];
    { std::lock_guard<std::mutex> g(w.mu); w.jobs.push_back(std::move(job)); }
    w.cv.notify_one();
  }

 private:
  struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool stop = false;
    std::thread th;
  };
  void run(size_t i) {
    Worker &w = w_[i];
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.stop || !w.jobs.empty(); });
        if (w.jobs.empty()) return;
        job = std::move(w.jobs.front());
        w.jobs.pop_front();
      }
      job();
    }
  }
  std::deque<Worker> w_;   // (deque: Worker is neither movable nor copyable)
};

// ---- RCCL (option shard-gather = 1) -----------------------------------------------------------------------------------
// north_star names "RCCL all-gather of per-shard top-k over xGMI".  The per-shard lists are 30 KiB per shard at B = 256,
// k = 10 -- latency-bound -- so the default gather is one-shot peer copies into the serving device (above); the collective
// is selectable beside it so that a hardware run can A/B the two (bench.py --gpus N prints both).  The library does not
// link librccl: it is loaded on first use -- /opt/rocm's copy first, the one built against the HIP runtime this library
// links (a process that also holds torch has a second RCCL, tied to torch's own HIP runtime) -- and a failure to load or
// initialise it FAILS the search that asked for it; there is no silent return to the peer copies.
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::vector<ncclComm_t> comms;   // one per DISTINCT device of the index, rank = the device's group index (ShardLayout)

  Status load() {
    if (lib) return Status::Ok();
    if (vk_test_rccl_library) {
      const char *name = vk_test_rccl_library();
      lib = name ? dlopen(name, RTLD_NOW | RTLD_LOCAL) : dlopen(nullptr, RTLD_NOW);   // nullptr: the program itself
    } else {
      for (const char *name : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
      }
    }
    if (!lib) return Status::Err(VK_ERR_INTERNAL, std::string("shard-gather = 1 needs RCCL: ") + dlerror());
    auto sym = [&](const char *n) { return dlsym(lib, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString)
      return Status::Err(VK_ERR_INTERNAL, "shard-gather = 1: librccl lacks a symbol the gather needs");
    return Status::Ok();
  }
  Status check(ncclResult_t r, const char *what) const {
    if (r == ncclSuccess) return Status::Ok();
    return Status::Err(VK_ERR_INTERNAL, std::string("RCCL ") + what + ": " + GetErrorString(r));
  }
  ~Rccl() {
    for (ncclComm_t c : comms)
      if (c) (void)CommDestroy(c);
    // (the library stays loaded: other indexes of the process may hold communicators of it)
  }
};

}  // namespace

class ShardedIndex final : public Index {
 public:
  ShardedIndex(const vk_index_params &p, std::vector<int> devices)
      : Index(p), devices_(std::move(devices)), counts_(devices_.size(), 0), capacity_(p.initial_cap) {}

  // an option applies to the sharded index and to every shard (kernel selection and scratch budgets live in the shards)
  Status set_option(const char *name, uint64_t value) override {
    VK_TRY(opt_.set(name, value));
    for (auto &s : shards_) VK_TRY(s->set_option(name, value));
    return Status::Ok();
  }

  Status init() {
    const size_t S = devices_.size();
    for (size_t s = 0; s < S; ++s) {
      vk_index_params sp = params_;
      sp.n_shards = 0;
      sp.device_id = devices_[s];
      // a shard starts with its share of the capacity and grows on demand (the limit the caller sees is capacity_)
      sp.initial_cap = std::max<uint64_t>(1024, (params_.initial_cap + S - 1) / S);
      std::unique_ptr<Index> sub;
      if (params_.algo == VK_ALGO_FLAT) VK_TRY(create_flat(sp, &sub));
      else VK_TRY(create_hnsw(sp, &sub));
      shards_.push_back(std::move(sub));
      shard_cap_.push_back(sp.initial_cap);
    }
    // One enqueue thread per DEVICE, not per shard: launches into one device go through one runtime lock, and threads that
    // share it only queue up behind each other (8 logical shards on one GPU: 330 us enqueued one after the other from one
    // thread, 230-620 us from eight threads); across devices the enqueueing does run side by side.  The caller's thread
    // takes the serving device's shards.
    const bool threads_on = opt_.get(kOptShardThreads) != 0;
    lay_ = ShardLayout::from_devices(devices_);   // (shard_layout.hpp: groups by device, the serving device's first)
    if (lay_.G() > 1 && threads_on) workers_ = std::make_unique<ShardWorkers>(lay_.G() - 1);
    // Peer access between the devices involved.  The fan-out's broadcast and gather are peer copies over xGMI; without peer
    // access the runtime would stage every one of them through host memory -- an index that LOOKS multi-GPU and runs at PCIe
    // latency.  That is refused here, loudly, rather than discovered in production (the option shard-allow-staged lifts it
    // for boxes whose devices really have no direct path).
    for (size_t a = 0; a < S; ++a)
      for (size_t b = 0; b < S; ++b)
        if (devices_[a] != devices_[b]) {
          int can = 0;
          const hipError_t ce = hipDeviceCanAccessPeer(&can, devices_[a], devices_[b]);
          if (ce == hipSuccess && can) {
            (void)hipSetDevice(devices_[a]);
            hipError_t e = hipDeviceEnablePeerAccess(devices_[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
              return Status::Err(VK_ERR_INTERNAL, "hipDeviceEnablePeerAccess(" + std::to_string(devices_[a]) + " -> " + std::to_string(devices_[b]) +
                                                      "): " + hipGetErrorString(e));
            (void)hipGetLastError();
          } else if (opt_.get(kOptShardAllowStaged) == 0) {
            return Status::Err(VK_ERR_NO_DEVICE, "device " + std::to_string(devices_[a]) + " has no peer access to device " + std::to_string(devices_[b]) +
                                                     ": the sharded index would stage every broadcast and gather through host memory "
                                                     "(set VK_SHARD_ALLOW_STAGED=1 / option shard-allow-staged to accept that)");
          }
        }
    return Status::Ok();
  }

  uint32_t shard_count() const override { return (uint32_t)shards_.size(); }
  void filter_devices(std::vector<int> *out) const override { *out = devices_; }   // (a copy per distinct device)

  // ---- mutations ------------------------------------------------------------------------------------
  Status add(uint64_t label, const float *row) override {
    uint32_t s;
    bool fresh;
    VK_TRY(route_for_add(label, &s, &fresh));
    Status st = add_to_shard(s, label, row);
    if (!st.ok() && fresh) unroute(label, s);
    return st;
  }

  Status add_batch(const uint64_t *labels, const float *rows, uint64_t n) override {
    const size_t S = shards_.size();
    std::vector<uint64_t> iota;
    if (!labels) {
      iota.resize(n);
      for (uint64_t i = 0; i < n; ++i) iota[i] = i;
      labels = iota.data();
    }
    // shard of every row: a known label stays where it is, new labels are dealt out in contiguous runs that even out
    // the shard sizes (an all-new bulk load of N rows: rows [s*N/S, (s+1)*N/S) to shard s)
    std::vector<uint32_t> shard_of(n);
    std::vector<uint8_t> fresh_row(n, 0);   // rows whose label got its route in this call (rolled back if the shard fails)
    bool over_capacity = false;
    uint64_t n_used = n;
    {
      std::unique_lock<std::shared_mutex> lk(rw_);
      uint64_t total = 0;
      for (uint64_t c : counts_) total += c;
      // (one hash operation per row: a new label gets its route entry at once, with a placeholder for the shard -- so that the
      //  same label again later in the batch is recognised and counted once; r06 found the capacity check counting it twice)
      std::vector<uint64_t> pending;        // rows whose label is new in this call, first occurrences and repeats, in order
      std::vector<uint32_t *> slot;         // ... and the route entry of each (node-based map: the address is stable)
      uint64_t n_fresh = 0;
      for (uint64_t i = 0; i < n; ++i) {
        auto ins = route_.emplace(labels[i], kMaxShards);
        if (!ins.second && ins.first->second != kMaxShards) { shard_of[i] = ins.first->second; continue; }
        if (ins.second) {
          if (total + n_fresh >= capacity_) {   // addPoint fails at the limit; everything before it is in
            route_.erase(ins.first);
            over_capacity = true;
            n_used = i;
            break;
          }
          ++n_fresh;
        }
        shard_of[i] = kMaxShards;   // marks "new"
        pending.push_back(i);
        slot.push_back(&ins.first->second);
      }
      const uint64_t after = total + n_fresh;
      size_t s = 0;
      uint64_t quota = 0;
      auto next_quota = [&]() {
        for (; s < S; ++s) {
          const uint64_t target = (after * (s + 1)) / S - (after * s) / S;   // even split of the final size
          if (target > counts_[s]) { quota = target - counts_[s]; return; }
        }
        quota = ~0ull;
        s = S - 1;
      };
      next_quota();
      for (size_t t = 0; t < pending.size(); ++t) {
        const uint64_t i = pending[t];
        uint32_t &r = *slot[t];
        if (r == kMaxShards) {             // its first occurrence: dealt out; a repeat follows the first
          if (quota == 0) { ++s; next_quota(); }
          r = (uint32_t)s;
          counts_[s]++;
          quota--;
          fresh_row[i] = 1;
        }
        shard_of[i] = r;
      }
    }
    // one add_batch per shard, concurrently (each shard has its own device, streams and -- HNSW -- graph builder)
    std::vector<Status> res(S);
    std::vector<std::thread> th;
    for (size_t s = 0; s < S; ++s) {
      th.emplace_back([&, s]() {
        // rows of shard s: one contiguous slice of the input in the usual case (no copy), else gathered
        uint64_t first = n_used, last = 0, cnt = 0;
        for (uint64_t i = 0; i < n_used; ++i)
          if (shard_of[i] == s) { first = std::min(first, i); last = i; ++cnt; }
        if (cnt == 0) return;
        const uint32_t dim = params_.dim;
        if (last - first + 1 == cnt) {
          res[s] = shard_add_batch((uint32_t)s, labels + first, rows + first * dim, cnt);
          return;
        }
        std::vector<uint64_t> ls;
        std::vector<float> rs;
        ls.reserve(cnt);
        rs.reserve(cnt * dim);
        for (uint64_t i = first; i <= last; ++i)
          if (shard_of[i] == s) {
            ls.push_back(labels[i]);
            rs.insert(rs.end(), rows + i * dim, rows + (i + 1) * dim);
          }
        res[s] = shard_add_batch((uint32_t)s, ls.data(), rs.data(), cnt);
      });
    }
    for (auto &t : th) t.join();
    // A shard that failed (out of memory, a resize that did not go through) may hold none, some or all of its rows: the
    // fresh labels it does NOT hold lose their route again, so that contains / get_row / distance do not resolve to a
    // shard without the row, the capacity accounting stays true and a retry does not meet its own leftovers.
    for (size_t s = 0; s < S; ++s) {
      if (res[s].ok()) continue;
      std::unique_lock<std::shared_mutex> lk(rw_);
      for (uint64_t i = 0; i < n_used; ++i) {
        if (shard_of[i] != s || !fresh_row[i]) continue;
        bool held = false;
        if (shards_[s]->contains(labels[i], &held).ok() && held) continue;
        auto it = route_.find(labels[i]);
        if (it != route_.end() && it->second == s) { route_.erase(it); counts_[s]--; }
      }
    }
    for (const Status &st : res)
      if (!st.ok()) return st;
    if (over_capacity) return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
    return Status::Ok();
  }

  Status remove(uint64_t label) override {
    uint32_t s;
    {
      std::unique_lock<std::shared_mutex> lk(rw_);
      auto it = route_.find(label);
      if (it == route_.end())   // FLAT ignores unknown labels (bruteforce.h:95-98), HNSW reports them
        return params_.algo == VK_ALGO_FLAT ? Status::Ok() : Status::Err(VK_ERR_NOT_FOUND, "Label not found");
      s = it->second;
      if (params_.algo == VK_ALGO_FLAT) {   // the row is gone; an HNSW tombstone keeps its slot (and its shard)
        route_.erase(it);
        counts_[s]--;
      }
    }
    return shards_[s]->remove(label);
  }

  Status resize(uint64_t new_max) override {
    std::unique_lock<std::shared_mutex> lk(rw_);
    capacity_ = new_max;
    return Status::Ok();
  }

  Status set_ef(uint32_t ef) override {
    for (auto &s : shards_) VK_TRY(s->set_ef(ef));
    return Status::Ok();
  }

  Status flush() override {
    for (auto &s : shards_) VK_TRY(s->flush());
    return Status::Ok();
  }

  // ---- queries --------------------------------------------------------------------------------------
  Status search(const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) override {
    if (rq.nq == 0) return Status::Ok();
    if (rq.query_tab) {   // (the dispatcher's batches: one pointer per query) -> one block, as the fan-out uploads it
      std::vector<float> Q((size_t)rq.nq * params_.dim);
      for (uint64_t q = 0; q < rq.nq; ++q) memcpy(Q.data() + q * params_.dim, rq.query_tab[q], (size_t)params_.dim * 4);
      SearchRequest g = rq;
      g.queries = Q.data();
      g.query_tab = nullptr;
      return search(g, out_dist, out_label, out_n);
    }
    // one filter per query: host bitmaps (and FLAT, whose kernels take one bitmap per pass) are served in runs of queries
    // that share a filter; device-resident filters of an HNSW index travel to the shards as they are, one launch per shard
    const bool tab_to_shards = rq.filter_tab && !rq.allow_tab && params_.algo == VK_ALGO_HNSW && flat_scan_slots_per_lane(rq.k) != 0;
    if ((rq.allow_tab || rq.filter_tab) && !tab_to_shards) return search_grouped_by_filter(this, rq, out_dist, out_label, out_n);
    if (rq.k == 0) {
      for (uint64_t q = 0; q < rq.nq; ++q) out_n[q] = 0;
      return Status::Ok();
    }
    if (cancel_raised(rq.cancel_flag) && !rq.partial_ok && params_.algo == VK_ALGO_HNSW)
      return Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
    if (flat_scan_slots_per_lane(rq.k) == 0) return search_by_host_merge(rq, out_dist, out_label, out_n);
    MultiLease lease(*this);
    MultiCtx *mc = lease.mc;
    // (whatever fails below, nothing of this call may still be in flight on the context's buffers when the lease ends)
    Status st = search_on(mc, rq, out_dist, out_label, out_n);
    if (!st.ok() && st.code != VK_ERR_CANCELLED) quiesce(mc, mc->s0);
    return st;
  }

  Status search_on(MultiCtx *mc, const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
    (void)hipSetDevice(mc->dev0);
    const uint32_t dim = params_.dim;
    const size_t qbytes = (size_t)rq.nq * dim * 4, nk = (size_t)rq.nq * rq.k;
    VK_TRY(mc->h_q.ensure(qbytes));
    VK_TRY(mc->d_q.ensure(qbytes));
    memcpy(mc->h_q.p, rq.queries, qbytes);
    VK_HIP_TRY(hipMemcpyAsync(mc->d_q.p, mc->h_q.p, qbytes, hipMemcpyHostToDevice, mc->s0));
    const uint64_t *d_allow = nullptr;
    if (rq.filter) {   // device-resident on every shard's device (filter_set.hpp): nothing to upload or broadcast
      if (!rq.filter->bits_on(mc->dev0)) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's devices");
    } else if (rq.allow_bits) {
      const size_t words = (size_t)((rq.allow_nbits + 63) / 64);
      VK_TRY(mc->d_allow.ensure(std::max<size_t>(words * 8, 8)));
      if (words) VK_HIP_TRY(hipMemcpyAsync(mc->d_allow.p, rq.allow_bits, words * 8, hipMemcpyHostToDevice, mc->s0));
      d_allow = mc->d_allow.as<uint64_t>();
    }
    const uint32_t *d_cancel = nullptr;
    if (rq.cancel_flag) {
      VK_TRY(mc->h_cancel.ensure(64));
      *mc->h_cancel.as<volatile uint32_t>() = cancel_raised(rq.cancel_flag) ? 1u : 0u;
      d_cancel = mc->h_cancel.as<uint32_t>();
    }
    VK_TRY(mc->d_fin_d.ensure(nk * 4));
    VK_TRY(mc->d_fin_l.ensure(nk * 8));
    VK_TRY(mc->d_fin_n.ensure(rq.nq * 4));
    VK_TRY(mc->h_fin_d.ensure(nk * 4));
    VK_TRY(mc->h_fin_l.ensure(nk * 8));
    VK_TRY(mc->h_fin_n.ensure(rq.nq * 4));
    SearchRequest drq = rq;
    drq.queries = mc->d_q.as<float>();
    drq.allow_bits = d_allow;
    if (rq.filter) drq.allow_nbits = rq.filter->nbits();
    drq.cancel_word = d_cancel;
    VK_TRY(fan_out(mc, drq, mc->d_fin_d.as<float>(), mc->d_fin_l.as<uint64_t>(), mc->d_fin_n.as<uint32_t>(), mc->s0));
    (void)hipSetDevice(mc->dev0);
    VK_HIP_TRY(hipMemcpyAsync(mc->h_fin_d.p, mc->d_fin_d.p, nk * 4, hipMemcpyDeviceToHost, mc->s0));
    VK_HIP_TRY(hipMemcpyAsync(mc->h_fin_l.p, mc->d_fin_l.p, nk * 8, hipMemcpyDeviceToHost, mc->s0));
    VK_HIP_TRY(hipMemcpyAsync(mc->h_fin_n.p, mc->d_fin_n.p, rq.nq * 4, hipMemcpyDeviceToHost, mc->s0));
    // wait; with a cancel flag, relay it to the word every shard's kernels poll
    if (!rq.cancel_flag) {
      VK_HIP_TRY(hipStreamSynchronize(mc->s0));
    } else {
      volatile uint32_t *word = mc->h_cancel.as<volatile uint32_t>();
      for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(mc->s0);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return Status::Err(VK_ERR_INTERNAL, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        if (cancel_raised(rq.cancel_flag)) *word = 1u;
        if (spins < 2000) __builtin_ia32_pause();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      if (cancel_raised(rq.cancel_flag) && !rq.partial_ok && params_.algo == VK_ALGO_HNSW)
        return Status::Err(VK_ERR_CANCELLED, "Search operation cancelled due to timeout");
    }
    for (uint64_t q = 0; q < rq.nq; ++q) {
      const uint32_t m = mc->h_fin_n.as<uint32_t>()[q];
      out_n[q] = m;
      memcpy(out_dist + q * rq.k, mc->h_fin_d.as<float>() + q * rq.k, (size_t)m * 4);
      memcpy(out_label + q * rq.k, mc->h_fin_l.as<uint64_t>() + q * rq.k, (size_t)m * 8);
    }
    return Status::Ok();
  }

  // queries / outputs on the serving device (the first shard's), work enqueued on `stream` without a host sync
  Status search_device(const SearchRequest &rq, float *d_out_dist, uint64_t *d_out_label, uint32_t *d_out_n,
                       hipStream_t stream) override {
    if (rq.nq == 0) return Status::Ok();
    if (rq.k == 0 || flat_scan_slots_per_lane(rq.k) == 0)
      return Status::Err(VK_ERR_INVALID, "search_batch_device on a sharded index needs 0 < k <= 1024");
    MultiLease lease(*this, stream);
    MultiCtx *mc = lease.mc;
    hipStream_t s0 = stream ? stream : mc->s0;
    Status st = fan_out(mc, rq, d_out_dist, d_out_label, d_out_n, s0);
    (void)hipSetDevice(mc->dev0);
    if (hipEventRecord(mc->busy, s0) == hipSuccess) mc->has_busy = true;
    return st;
  }

  Status label_distances(const float *query, const uint64_t *labels, uint64_t n, float *out_dist, uint8_t *found) override {
    const size_t S = shards_.size();
    std::vector<std::vector<uint64_t>> idx(S), lab(S);
    {
      std::shared_lock<std::shared_mutex> lk(rw_);
      for (uint64_t i = 0; i < n; ++i) {
        found[i] = 0;
        auto it = route_.find(labels[i]);
        if (it == route_.end()) continue;
        idx[it->second].push_back(i);
        lab[it->second].push_back(labels[i]);
      }
    }
    for (size_t s = 0; s < S; ++s) {
      if (lab[s].empty()) continue;
      std::vector<float> d(lab[s].size());
      std::vector<uint8_t> f(lab[s].size());
      VK_TRY(shards_[s]->label_distances(query, lab[s].data(), lab[s].size(), d.data(), f.data()));
      for (size_t j = 0; j < lab[s].size(); ++j) {
        found[idx[s][j]] = f[j];
        out_dist[idx[s][j]] = d[j];
      }
    }
    return Status::Ok();
  }

  Status distance(uint64_t label, const float *query, float *out) override {
    uint32_t s;
    if (!route_of(label, &s)) return Status::Err(VK_ERR_NOT_FOUND, "Couldn't find internal id");
    return shards_[s]->distance(label, query, out);
  }
  Status get_row(uint64_t label, float *out) override {
    uint32_t s;
    if (!route_of(label, &s)) return Status::Err(VK_ERR_NOT_FOUND, "label not found");
    return shards_[s]->get_row(label, out);
  }
  Status contains(uint64_t label, bool *found) override {
    uint32_t s;
    *found = false;
    if (!route_of(label, &s)) return Status::Ok();
    return shards_[s]->contains(label, found);
  }

  Status stats(vk_index_stats *out) override {
    memset(out, 0, sizeof(*out));
    out->max_level = -1;
    for (size_t s = 0; s < shards_.size(); ++s) {
      vk_index_stats t;
      VK_TRY(shards_[s]->stats(&t));
      out->count += t.count;
      out->deleted += t.deleted;
      out->device_bytes += t.device_bytes;
      out->host_bytes += t.host_bytes;
      out->staged_ops += t.staged_ops;
      out->max_level = std::max(out->max_level, t.max_level);
      if (s == 0) out->entry_point = t.entry_point;
      // candidate-filter timing of a fan-out: the shards run side by side, the step waits for the slowest
      out->filter_batches = s == 0 ? t.filter_batches : std::min(out->filter_batches, t.filter_batches);
      out->filter_kernel_ns = std::max(out->filter_kernel_ns, t.filter_kernel_ns);
      out->last_n_eval += t.last_n_eval;
      out->last_n_hops += t.last_n_hops;
      out->total_n_eval += t.total_n_eval;
      out->total_n_hops += t.total_n_hops;
      out->tombstoned_bytes += t.tombstoned_bytes;
      out->max_label = std::max(out->max_label, t.max_label);
      out->staged_adds += t.staged_adds;
      out->staged_adds_device += t.staged_adds_device;
      out->last_filter_final_rows += t.last_filter_final_rows;
    }
    out->fanout_calls = fanout_calls_.load(std::memory_order_relaxed);
    out->fanout_enqueue_ns = fanout_ns_.load(std::memory_order_relaxed);
    out->rccl_gathers = rccl_gathers_.load(std::memory_order_relaxed);
    std::shared_lock<std::shared_mutex> lk(rw_);
    out->capacity = capacity_;
    out->host_bytes += route_.size() * 24;
    return Status::Ok();
  }

  // one shard's own statistics (bench.py: the slowest shard's kernel time per launch is max over shards of a shard's
  // delta of filter_kernel_ns over its delta of filter_batches -- not a difference of maxima)
  Status shard_stats(uint32_t s, vk_index_stats *out) override {
    if (s >= shards_.size()) return Status::Err(VK_ERR_INVALID, "shard out of range");
    return shards_[s]->stats(out);
  }

  Status device_rows(uint64_t, void **, uint64_t *) override {
    return Status::Err(VK_ERR_INVALID, "a sharded index loads device rows shard by shard (vk_index_shard_device_rows)");
  }
  Status commit_device_rows(uint64_t, const uint64_t *) override {
    return Status::Err(VK_ERR_INVALID, "a sharded index loads device rows shard by shard (vk_index_shard_commit_device_rows)");
  }
  Status shard_device_rows(uint32_t s, uint64_t n, void **d_rows, uint64_t *stride) override {
    if (s >= shards_.size()) return Status::Err(VK_ERR_INVALID, "shard out of range");
    VK_TRY(shards_[s]->resize(std::max<uint64_t>(n, shard_cap_[s])));
    shard_cap_[s] = std::max<uint64_t>(n, shard_cap_[s]);
    return shards_[s]->device_rows(n, d_rows, stride);
  }
  Status shard_commit_device_rows(uint32_t s, uint64_t n, const uint64_t *labels) override {
    if (s >= shards_.size()) return Status::Err(VK_ERR_INVALID, "shard out of range");
    if (!labels) return Status::Err(VK_ERR_INVALID, "a sharded bulk load needs explicit labels (unique across the shards)");
    {
      std::unique_lock<std::shared_mutex> lk(rw_);
      uint64_t total = 0;
      for (uint64_t c : counts_) total += c;
      if (total + n > capacity_) return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
      // (emplace, not count-then-insert: a label twice inside `labels` is a duplicate too, and counts_ must not count it)
      route_.reserve(route_.size() + n);
      for (uint64_t i = 0; i < n; ++i)
        if (!route_.emplace(labels[i], s).second) {
          for (uint64_t j = 0; j < i; ++j) route_.erase(labels[j]);
          return Status::Err(VK_ERR_INVALID, "duplicate label in bulk load");
        }
      counts_[s] += n;
    }
    Status st = shards_[s]->commit_device_rows(n, labels);
    if (!st.ok()) {   // the shard holds none of them (commit_device_rows is all-or-nothing): take the routes back
      std::unique_lock<std::shared_mutex> lk(rw_);
      for (uint64_t i = 0; i < n; ++i) route_.erase(labels[i]);
      counts_[s] -= n;
    }
    return st;
  }

  Status save(vk_write_chunk_fn fn, void *user) override;
  Status load_from(vk_read_chunk_fn fn, void *user);

 private:
  // ---- routing ------------------------------------------------------------------------------------------
  bool route_of(uint64_t label, uint32_t *s) {
    std::shared_lock<std::shared_mutex> lk(rw_);
    auto it = route_.find(label);
    if (it == route_.end()) return false;
    *s = it->second;
    return true;
  }
  Status route_for_add(uint64_t label, uint32_t *s, bool *fresh) {
    std::unique_lock<std::shared_mutex> lk(rw_);
    auto it = route_.find(label);
    if (it != route_.end()) { *s = it->second; *fresh = false; return Status::Ok(); }
    uint64_t total = 0;
    for (uint64_t c : counts_) total += c;
    if (total >= capacity_) return Status::Err(VK_ERR_CAPACITY, "The number of elements exceeds the specified limit");
    uint32_t best = 0;
    for (uint32_t i = 1; i < counts_.size(); ++i)
      if (counts_[i] < counts_[best]) best = i;
    route_.emplace(label, best);
    counts_[best]++;
    *s = best;
    *fresh = true;
    return Status::Ok();
  }
  void unroute(uint64_t label, uint32_t s) {
    std::unique_lock<std::shared_mutex> lk(rw_);
    if (route_.erase(label)) counts_[s]--;
  }
  // a shard's own capacity is an internal matter: grow it and retry (the caller's limit is capacity_)
  Status add_to_shard(uint32_t s, uint64_t label, const float *row) {
    Status st = shards_[s]->add(label, row);
    if (st.code != VK_ERR_CAPACITY) return st;
    VK_TRY(grow_shard(s, 1));
    return shards_[s]->add(label, row);
  }
  Status shard_add_batch(uint32_t s, const uint64_t *labels, const float *rows, uint64_t n) {
    vk_index_stats t;
    VK_TRY(shards_[s]->stats(&t));
    if (t.count + n > shard_cap_[s]) VK_TRY(grow_shard(s, t.count + n - shard_cap_[s]));
    return shards_[s]->add_batch(labels, rows, n);
  }
  Status grow_shard(uint32_t s, uint64_t at_least) {
    std::lock_guard<std::mutex> g(grow_mu_);
    const uint64_t want = std::max<uint64_t>(shard_cap_[s] + at_least, shard_cap_[s] + shard_cap_[s] / 2);
    VK_TRY(shards_[s]->resize(want));
    shard_cap_[s] = want;
    return Status::Ok();
  }

  // ---- per-call contexts ---------------------------------------------------------------------------------
  MultiCtx *acquire(hipStream_t on) {
    MultiCtx *mc = nullptr;
    {
      std::unique_lock<std::mutex> lk(ctx_mu_);
      for (;;) {
        if (!free_.empty()) { mc = free_.back(); free_.pop_back(); break; }
        if (all_.size() < 8) {
          auto n = std::make_unique<MultiCtx>();
          n->dev0 = devices_[0];
          (void)hipSetDevice(n->dev0);
          (void)hipStreamCreateWithFlags(&n->s0, hipStreamNonBlocking);
          (void)hipEventCreateWithFlags(&n->ready, hipEventDisableTiming);
          (void)hipEventCreateWithFlags(&n->busy, hipEventDisableTiming);
          n->lane.resize(devices_.size());
          for (size_t s = 0; s < devices_.size(); ++s) {
            ShardLane &l = n->lane[s];
            l.device = devices_[s];
            (void)hipSetDevice(l.device);
            (void)hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
            (void)hipEventCreateWithFlags(&l.done, hipEventDisableTiming);
          }
          n->group.resize(lay_.G());
          for (size_t gi = 0; gi < lay_.G(); ++gi) {
            n->group[gi].device = lay_.group_device[gi];
            (void)hipSetDevice(n->group[gi].device);
            (void)hipEventCreateWithFlags(&n->group[gi].gathered, hipEventDisableTiming);
          }
          all_.push_back(std::move(n));
          mc = all_.back().get();
          break;
        }
        ctx_cv_.wait(lk);
      }
    }
    if (mc->has_busy) {   // its previous user's work may still be in flight: the new work goes behind it
      (void)hipSetDevice(mc->dev0);
      if (hipStreamWaitEvent(on ? on : mc->s0, mc->busy, 0) != hipSuccess) (void)hipEventSynchronize(mc->busy);
      mc->has_busy = false;
    }
    return mc;
  }
  void release(MultiCtx *mc) {
    {
      std::lock_guard<std::mutex> lk(ctx_mu_);
      free_.push_back(mc);
    }
    ctx_cv_.notify_one();
  }
  struct MultiLease {
    ShardedIndex &ix;
    MultiCtx *mc;
    explicit MultiLease(ShardedIndex &i, hipStream_t on = nullptr) : ix(i), mc(i.acquire(on)) {}
    ~MultiLease() { ix.release(mc); }
  };

  // One shard's part of a fan-out, enqueued on the shard's lane stream (any thread): wait for the queries, broadcast by
  // peer copy unless the shard lives on the serving device, search, send the lists to their slice of the gathered array.
  Status enqueue_shard(MultiCtx *mc, size_t s, const SearchRequest &rq, bool rccl) {
    const uint32_t dim = params_.dim;
    const size_t qbytes = (size_t)rq.nq * dim * 4, nk = (size_t)rq.nq * rq.k;
    const size_t abytes = rq.allow_bits ? (size_t)((rq.allow_nbits + 63) / 64) * 8 : 0;
    ShardLane &l = mc->lane[s];
    VK_HIP_TRY(hipSetDevice(l.device));
    VK_HIP_TRY(hipStreamWaitEvent(l.stream, mc->ready, 0));
    SearchRequest srq = rq;
    srq.cancel_flag = nullptr;
    srq.filter = nullptr;          // (filter_tab stays: the shard points its table at the copies on its own device)
    srq.member_cancel = nullptr;
    if (rq.filter) {
      srq.allow_bits = rq.filter->bits_on(l.device);
      if (!srq.allow_bits) return Status::Err(VK_ERR_INVALID, "the filter was not built for this index's devices");
    }
    const uint64_t ef_pct = opt_.get(kOptShardEfPct);
    if (params_.algo == VK_ALGO_HNSW && ef_pct != 100) {
      // per-shard ef policy (option shard-ef-pct, initialised from vk_index_params.shard_ef_pct): a fraction of the ef one
      // graph over all rows would be given
      uint64_t ef = rq.ef ? rq.ef : (params_.ef_runtime ? params_.ef_runtime : 10);
      ef = (ef * ef_pct + 99) / 100;
      srq.ef = std::max<uint64_t>(std::max<uint64_t>(ef, rq.k), 1);
    }
    float *od = mc->d_all_d.as<float>() + lay_.peer_slice(s, nk);
    uint64_t *ol = mc->d_all_l.as<uint64_t>() + lay_.peer_slice(s, nk);
    uint32_t *on = mc->d_all_n.as<uint32_t>() + s * rq.nq;
    const bool local = l.device == mc->dev0;
    if (!local) {   // broadcast by peer copy, answer into the shard's own buffers
      VK_TRY(l.d_q.ensure(qbytes));
      VK_HIP_TRY(hipMemcpyPeerAsync(l.d_q.p, l.device, rq.queries, mc->dev0, qbytes, l.stream));
      srq.queries = l.d_q.as<float>();
      if (rq.allow_bits && !rq.filter) {
        VK_TRY(l.d_allow.ensure(std::max<size_t>(abytes, 8)));
        if (abytes) VK_HIP_TRY(hipMemcpyPeerAsync(l.d_allow.p, l.device, rq.allow_bits, mc->dev0, abytes, l.stream));
        srq.allow_bits = l.d_allow.as<uint64_t>();
      }
      VK_TRY(l.d_out_d.ensure(nk * 4));
      VK_TRY(l.d_out_l.ensure(nk * 8));
      VK_TRY(l.d_out_n.ensure(rq.nq * 4));
      od = l.d_out_d.as<float>();
      ol = l.d_out_l.as<uint64_t>();
      on = l.d_out_n.as<uint32_t>();
    }
    if (rccl) {   // the lists go to this shard's slot of its DEVICE's send buffer; the all-gather moves them (fan_out)
      GatherGroup &g = mc->group[lay_.group_of[s]];
      od = g.send_d.as<float>() + lay_.send_slot(s, nk);
      ol = g.send_l.as<uint64_t>() + lay_.send_slot(s, nk);
    }
    VK_TRY(shards_[s]->search_device(srq, od, ol, on, l.stream));
    VK_HIP_TRY(hipSetDevice(l.device));
    if (!local && !rccl) {   // the shard's lists -> their slice of the gathered array on the serving device
      VK_HIP_TRY(hipMemcpyPeerAsync(mc->d_all_d.as<float>() + lay_.peer_slice(s, nk), mc->dev0, od, l.device, nk * 4, l.stream));
      VK_HIP_TRY(hipMemcpyPeerAsync(mc->d_all_l.as<uint64_t>() + lay_.peer_slice(s, nk), mc->dev0, ol, l.device, nk * 8, l.stream));
    }
    VK_HIP_TRY(hipEventRecord(l.done, l.stream));
    return Status::Ok();
  }

  // A fan-out that failed part-way leaves kernels and peer copies of the shards that WERE launched in flight on the
  // context's buffers: they are waited for before the context goes back to the pool (its next user would otherwise
  // reuse -- or DevBuf::ensure free -- memory they still write).
  void quiesce(MultiCtx *mc, hipStream_t s0) {
    for (ShardLane &l : mc->lane) {
      (void)hipSetDevice(l.device);
      (void)hipStreamSynchronize(l.stream);
    }
    (void)hipSetDevice(mc->dev0);
    (void)hipStreamSynchronize(s0);
  }

  // shard-gather = 1, before the shards are enqueued: the communicators (first use), every device's send / receive buffers,
  // and (+inf, no label) in the send slots no shard of that device writes
  Status rccl_prepare(MultiCtx *mc, const SearchRequest &rq, size_t nk, hipStream_t s0) {
    const size_t G = lay_.G();
    {
      std::lock_guard<std::mutex> lk(rccl_mu_);
      VK_TRY(rccl_.load());
      if (rccl_.comms.empty()) {
        std::vector<ncclComm_t> comms(G, nullptr);   // rank = group index: the receive buffer is laid out in that order
        VK_TRY(rccl_.check(rccl_.CommInitAll(comms.data(), (int)G, lay_.group_device.data()), "ncclCommInitAll"));
        rccl_.comms = std::move(comms);
      }
    }
    for (size_t gi = 0; gi < G; ++gi) {
      GatherGroup &g = mc->group[gi];
      VK_HIP_TRY(hipSetDevice(g.device));
      VK_TRY(g.send_d.ensure(lay_.send_entries(nk) * 4));
      VK_TRY(g.send_l.ensure(lay_.send_entries(nk) * 8));
      VK_TRY(g.recv_d.ensure(lay_.recv_entries(nk) * 4));
      VK_TRY(g.recv_l.ensure(lay_.recv_entries(nk) * 8));
      if (lay_.pad_begin(gi) < lay_.P && g.filled_nk != nk) {   // the pad slots, once per batch shape (on the stream that gathers)
        hipStream_t cs = mc->lane[lay_.collective_lane(gi)].stream;
        for (size_t p = lay_.pad_begin(gi); p < lay_.P; ++p)
          VK_HIP_TRY(launch_fill_empty(g.send_d.as<float>() + p * nk, g.send_l.as<uint64_t>() + p * nk, nullptr, (uint32_t)rq.nq, (uint32_t)rq.k, cs));
        g.filled_nk = nk;
      }
    }
    (void)s0;
    return Status::Ok();
  }
  // ... and after them: on every device the stream of its last shard waits for the device's other shards, then ONE group
  // call issues the all-gathers of all devices (distances, labels) from this thread
  Status rccl_all_gather(MultiCtx *mc, size_t nk) {
    const size_t G = lay_.G();
    for (size_t gi = 0; gi < G; ++gi) {
      VK_HIP_TRY(hipSetDevice(mc->group[gi].device));
      hipStream_t cs = mc->lane[lay_.collective_lane(gi)].stream;
      for (size_t i = 0; i + 1 < lay_.groups[gi].size(); ++i) VK_HIP_TRY(hipStreamWaitEvent(cs, mc->lane[lay_.groups[gi][i]].done, 0));
    }
    {
      std::lock_guard<std::mutex> lk(rccl_mu_);
      VK_TRY(rccl_.check(rccl_.GroupStart(), "ncclGroupStart"));
      Status st = Status::Ok();
      for (size_t gi = 0; gi < G && st.ok(); ++gi) {
        GatherGroup &g = mc->group[gi];
        hipStream_t cs = mc->lane[lay_.collective_lane(gi)].stream;
        st = rccl_.check(rccl_.AllGather(g.send_d.p, g.recv_d.p, lay_.send_entries(nk), ncclFloat32, rccl_.comms[gi], cs), "ncclAllGather(distances)");
        if (st.ok()) st = rccl_.check(rccl_.AllGather(g.send_l.p, g.recv_l.p, lay_.send_entries(nk), ncclUint64, rccl_.comms[gi], cs), "ncclAllGather(labels)");
      }
      Status en = rccl_.check(rccl_.GroupEnd(), "ncclGroupEnd");
      VK_TRY(st);
      VK_TRY(en);
    }
    for (size_t gi = 0; gi < G; ++gi) {
      VK_HIP_TRY(hipSetDevice(mc->group[gi].device));
      hipStream_t cs = mc->lane[lay_.collective_lane(gi)].stream;
      VK_HIP_TRY(hipEventRecord(mc->group[gi].gathered, cs));
      // (the lane's `done` event is what quiesce / the next user of the context wait for: move it behind the collective)
      VK_HIP_TRY(hipEventRecord(mc->lane[lay_.collective_lane(gi)].done, cs));
    }
    rccl_gathers_.fetch_add(1, std::memory_order_relaxed);
    return Status::Ok();
  }

  // the fan-out itself: rq holds DEVICE pointers on the serving device, valid on stream s0; the merged answer is written
  // to d_out_* (serving device) by work enqueued on s0
  Status fan_out(MultiCtx *mc, const SearchRequest &rq, float *d_out_dist, uint64_t *d_out_label, uint32_t *d_out_n,
                 hipStream_t s0) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t S = shards_.size();
    const size_t nk = (size_t)rq.nq * rq.k;
    (void)hipSetDevice(mc->dev0);
    VK_TRY(mc->d_all_d.ensure(S * nk * 4));
    VK_TRY(mc->d_all_l.ensure(S * nk * 8));
    VK_TRY(mc->d_all_n.ensure(S * rq.nq * 4));
    const bool rccl = opt_.get(kOptShardGather) != 0;
    if (rccl) {
      Status st = rccl_prepare(mc, rq, nk, s0);
      if (!st.ok()) { quiesce(mc, s0); return st; }
      (void)hipSetDevice(mc->dev0);
    }
    VK_HIP_TRY(hipEventRecord(mc->ready, s0));   // queries (and filter) are in place on the serving device
    std::vector<Status> res(S);
    if (workers_) {
      ShardWorkers::Latch latch;
      latch.left = (uint32_t)(lay_.G() - 1);
      auto run_group = [this, mc, &rq, &res, rccl](size_t gi) {
        for (size_t s : lay_.groups[gi]) {
          try {
            res[s] = enqueue_shard(mc, s, rq, rccl);
          } catch (const std::exception &e) {
            res[s] = Status::Err(VK_ERR_INTERNAL, e.what());
          }
          if (!res[s].ok()) break;
        }
      };
      for (size_t gi = 1; gi < lay_.G(); ++gi)
        workers_->post(gi - 1, [gi, &run_group, &latch] {
          run_group(gi);
          latch.done();
        });
      run_group(0);
      latch.wait();
    } else {
      for (size_t s = 0; s < S; ++s) {
        res[s] = enqueue_shard(mc, s, rq, rccl);
        if (!res[s].ok()) break;
      }
    }
    for (const Status &st : res)
      if (!st.ok()) {
        quiesce(mc, s0);
        return st;
      }
    (void)hipSetDevice(mc->dev0);
    Status st = [&]() -> Status {
      MergeArgs m{};
      if (rccl) {
        VK_TRY(rccl_all_gather(mc, nk));
        (void)hipSetDevice(mc->dev0);
        VK_HIP_TRY(hipStreamWaitEvent(s0, mc->group[0].gathered, 0));
        m.in_dist = mc->group[0].recv_d.as<float>();
        m.in_label = mc->group[0].recv_l.as<uint64_t>();
        m.parts = (uint32_t)lay_.rccl_parts();
      } else {
        for (size_t s = 0; s < S; ++s) VK_HIP_TRY(hipStreamWaitEvent(s0, mc->lane[s].done, 0));
        m.in_dist = mc->d_all_d.as<float>();
        m.in_label = mc->d_all_l.as<uint64_t>();
        m.parts = (uint32_t)lay_.peer_parts();
      }
      m.part_stride = nk;
      m.q_stride = rq.k;
      m.per_part = (uint32_t)rq.k;
      m.k = (uint32_t)rq.k;
      m.out_ld = (uint32_t)rq.k;
      m.out_dist = d_out_dist;
      m.out_label = d_out_label;
      m.out_n = d_out_n;
      VK_HIP_TRY(launch_merge_topk(m, flat_scan_slots_per_lane(rq.k), rq.nq, s0));
      return Status::Ok();
    }();
    if (!st.ok()) {
      quiesce(mc, s0);
      return st;
    }
    fanout_calls_.fetch_add(1, std::memory_order_relaxed);
    fanout_ns_.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                         std::memory_order_relaxed);
    return Status::Ok();
  }

  // k beyond the device merge (FLAT pages such results on the host anyway): every shard's host answer, merged here
  Status search_by_host_merge(const SearchRequest &rq, float *out_dist, uint64_t *out_label, uint64_t *out_n) {
    const size_t S = shards_.size();
    const size_t nk = (size_t)rq.nq * rq.k;
    std::vector<float> d(S * nk);
    std::vector<uint64_t> l(S * nk), cnt(S * rq.nq);
    for (size_t s = 0; s < S; ++s)
      VK_TRY(shards_[s]->search(rq, d.data() + s * nk, l.data() + s * nk, cnt.data() + s * rq.nq));
    std::vector<std::pair<float, uint64_t>> all;
    for (uint64_t q = 0; q < rq.nq; ++q) {
      all.clear();
      for (size_t s = 0; s < S; ++s)
        for (uint64_t i = 0; i < cnt[s * rq.nq + q]; ++i)
          all.emplace_back(d[s * nk + q * rq.k + i], l[s * nk + q * rq.k + i]);
      const size_t m = std::min<size_t>(all.size(), rq.k);
      std::partial_sort(all.begin(), all.begin() + m, all.end());
      out_n[q] = m;
      for (size_t i = 0; i < m; ++i) { out_dist[q * rq.k + i] = all[i].first; out_label[q * rq.k + i] = all[i].second; }
    }
    return Status::Ok();
  }

  std::vector<int> devices_;
  std::unique_ptr<ShardWorkers> workers_;
  ShardLayout lay_;                                        // shards by device, send slots, receive offsets (shard_layout.hpp)
  Rccl rccl_;                                              // shard-gather = 1 (loaded and initialised on first use, under rccl_mu_)
  std::mutex rccl_mu_;                                     // collectives of one communicator set are issued by one thread at a time
  std::atomic<uint64_t> rccl_gathers_{0};
  std::atomic<uint64_t> fanout_calls_{0}, fanout_ns_{0};   // host time of fan_out (enqueue only), for vk_index_stats
  std::vector<std::unique_ptr<Index>> shards_;
  std::vector<uint64_t> shard_cap_;
  std::shared_mutex rw_;                              // route_, counts_, capacity_
  std::unordered_map<uint64_t, uint32_t> route_;      // label -> shard
  std::vector<uint64_t> counts_;                      // labels routed to each shard
  uint64_t capacity_;
  std::mutex grow_mu_;
  std::mutex ctx_mu_;
  std::condition_variable ctx_cv_;
  std::vector<MultiCtx *> free_;
  std::vector<std::unique_ptr<MultiCtx>> all_;
};

// ---- persistence ---------------------------------------------------------------------------------------------
// FLAT: ONE stream in the reference's layout (bruteforce.h:147-207): a header with the totals, then every shard's
// element chunks -- any BruteforceSearch::LoadIndex reads it, and any shard count loads it back.
// HNSW: one graph per shard cannot be one hnswlib stream; the shards' own streams follow a marker chunk
// ("VKSHARDS", shard count).  A plain single-graph stream loads too: its rows are re-inserted.
namespace {
struct SkipHeader {
  vk_write_chunk_fn fn;
  void *user;
  bool first = true;
};
int skip_header_cb(void *u, const void *data, uint64_t len) {
  SkipHeader *s = static_cast<SkipHeader *>(u);
  if (s->first) { s->first = false; return 0; }
  return s->fn(s->user, data, len);
}
}  // namespace

Status ShardedIndex::save(vk_write_chunk_fn fn, void *user) {
  if (params_.algo == VK_ALGO_FLAT) {
    vk_index_stats t;
    VK_TRY(stats(&t));
    std::string hdr;
    pb_put_varint_field(hdr, 1, t.capacity);
    pb_put_varint_field(hdr, 2, (uint64_t)params_.dim * 4 + 8);
    pb_put_varint_field(hdr, 3, t.count);
    if (fn(user, hdr.data(), hdr.size())) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
    for (auto &s : shards_) {
      SkipHeader sh{fn, user};
      VK_TRY(s->save(skip_header_cb, &sh));
    }
    return Status::Ok();
  }
  char marker[16] = {0};
  memcpy(marker, kShardMagic, 8);
  const uint64_t S = shards_.size();
  memcpy(marker + 8, &S, 8);
  if (fn(user, marker, sizeof marker)) return Status::Err(VK_ERR_INTERNAL, "write_chunk failed");
  for (auto &s : shards_) VK_TRY(s->save(fn, user));
  return Status::Ok();
}

Status ShardedIndex::load_from(vk_read_chunk_fn fn, void *user) {
  const uint32_t dim = params_.dim;
  const size_t vec = (size_t)dim * 4;
  std::vector<char> buf(std::max<size_t>(vec + 8 + 4 + 8 * 10000 + 4096, 1 << 16));
  uint64_t len = 0;
  if (fn(user, buf.data(), buf.size(), &len)) return Status::Err(VK_ERR_INTERNAL, "read_chunk failed");
  if (params_.algo == VK_ALGO_HNSW && len == 16 && memcmp(buf.data(), kShardMagic, 8) == 0) {
    uint64_t S;
    memcpy(&S, buf.data() + 8, 8);
    if (S != shards_.size()) return Status::Err(VK_ERR_INVALID, "the stream holds a different number of HNSW shards than the index definition");
    for (size_t s = 0; s < shards_.size(); ++s) {
      vk_index_params sp = params_;
      sp.n_shards = 0;
      sp.device_id = devices_[s];
      sp.initial_cap = shard_cap_[s];
      std::unique_ptr<Index> sub;
      VK_TRY(load_hnsw(sp, fn, user, &sub));
      shards_[s] = std::move(sub);
      // rebuild the routes from what the shard holds: its saved labels come back through a save into a collector
      struct Collect { ShardedIndex *self; uint32_t shard; size_t vec, sl0; uint64_t n = 0, seen = 0; bool hdr = true; } c{this, (uint32_t)s, vec, 0};
      vk_index_stats t;
      VK_TRY(shards_[s]->stats(&t));
      c.n = t.count;
      c.sl0 = ((size_t)params_.m * 2 + 1) * 4;
      auto cb = [](void *u, const void *data, uint64_t l) -> int {
        Collect *c = static_cast<Collect *>(u);
        if (c->hdr) { c->hdr = false; return 0; }
        if (c->seen < c->n && l == c->sl0 + c->vec + 8) {
          uint64_t lab;
          memcpy(&lab, static_cast<const char *>(data) + c->sl0 + c->vec, 8);
          if (c->self->route_.emplace(lab, c->shard).second) c->self->counts_[c->shard]++;
          c->seen++;
        }
        return 0;
      };
      VK_TRY(shards_[s]->save(cb, &c));
      shard_cap_[s] = std::max<uint64_t>(shard_cap_[s], t.capacity);
    }
    return Status::Ok();
  }
  // a single stream: FLAT elements (or the rows of one HNSW graph) dealt out to the shards
  uint64_t f[16] = {0};
  {
    PbReader r{reinterpret_cast<const uint8_t *>(buf.data()), reinterpret_cast<const uint8_t *>(buf.data()) + len};
    uint32_t field, wire;
    uint64_t val;
    while (r.next(&field, &wire, &val))
      if (field < 16 && wire == 0) f[field] = val;
  }
  const bool hnsw = params_.algo == VK_ALGO_HNSW;
  const uint64_t count = f[3];
  const size_t sl0 = hnsw ? ((size_t)params_.m * 2 + 1) * 4 : 0;
  const size_t elem = sl0 + vec + 8;
  if (!hnsw && f[2] != vec + 8) return Status::Err(VK_ERR_INTERNAL, "Persisted size_per_element does not match expectation.");
  if (hnsw && f[4] != elem) return Status::Err(VK_ERR_INTERNAL, "HNSW index load validation failed: serialized element size is inconsistent with the geometry");
  capacity_ = std::max<uint64_t>(capacity_, std::max<uint64_t>(hnsw ? f[2] : f[1], count));
  if (buf.size() < elem) buf.resize(elem);
  const uint64_t group = std::max<uint64_t>(1, ((uint64_t)64 << 20) / vec);
  std::vector<float> rows;
  std::vector<uint64_t> labs;
  auto flush_group = [&]() -> Status {
    if (labs.empty()) return Status::Ok();
    Status st = add_batch(labs.data(), rows.data(), labs.size());
    rows.clear();
    labs.clear();
    return st;
  };
  for (uint64_t i = 0; i < count; ++i) {
    if (fn(user, buf.data(), buf.size(), &len) || len != elem) return Status::Err(VK_ERR_INTERNAL, "truncated element chunk");
    if (hnsw && (reinterpret_cast<const uint32_t *>(buf.data())[0] & 0x00010000u)) continue;   // tombstoned: not re-inserted
    uint64_t lab;
    memcpy(&lab, buf.data() + sl0 + vec, 8);
    labs.push_back(lab);
    VK_TRY(observe_loaded_row(lab, buf.data() + sl0));
    const float *v = reinterpret_cast<const float *>(buf.data() + sl0);
    rows.insert(rows.end(), v, v + dim);
    if (labs.size() >= group) VK_TRY(flush_group());
  }
  VK_TRY(flush_group());
  if (hnsw) {   // drain the upper-level section of the graph that is not taken over
    for (uint64_t i = 0; i < count; ++i) {
      if (fn(user, buf.data(), buf.size(), &len) || len != 8) return Status::Err(VK_ERR_INTERNAL, "HNSW index load validation failed: link-list size chunk has the wrong size");
      uint64_t sz;
      memcpy(&sz, buf.data(), 8);
      if (!sz) continue;
      if (buf.size() < sz) buf.resize(sz);
      if (fn(user, buf.data(), buf.size(), &len) || len != sz) return Status::Err(VK_ERR_INTERNAL, "HNSW index load validation failed: upper-level link-list chunk has the wrong size");
    }
  }
  return flush();
}

static Status shard_devices(const vk_index_params &p, std::vector<int> *devices) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return Status::Err(VK_ERR_NO_DEVICE, "no HIP device: libvkindex needs a gfx950 GPU (no CPU fallback)");
  if (p.n_shards > kMaxShards) return Status::Err(VK_ERR_INVALID, "n_shards out of range");
  for (uint32_t s = 0; s < p.n_shards; ++s) {
    int d = p.shard_devices[s];
    if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
    if (d >= n) return Status::Err(VK_ERR_INVALID, "shard_devices: device out of range");
    devices->push_back(d);
  }
  return Status::Ok();
}

Status create_sharded(const vk_index_params &p, std::unique_ptr<Index> *out) {
  std::vector<int> devices;
  VK_TRY(shard_devices(p, &devices));
  auto ix = std::make_unique<ShardedIndex>(p, devices);
  VK_TRY(ix->init());
  *out = std::move(ix);
  return Status::Ok();
}

Status load_sharded(const vk_index_params &p, vk_read_chunk_fn fn, void *user, std::unique_ptr<Index> *out) {
  std::vector<int> devices;
  VK_TRY(shard_devices(p, &devices));
  auto ix = std::make_unique<ShardedIndex>(p, devices);
  VK_TRY(ix->init());
  VK_TRY(ix->load_from(fn, user));
  *out = std::move(ix);
  return Status::Ok();
}

}  // namespace vk
