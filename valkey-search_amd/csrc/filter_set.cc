// filter_set.cc -- see filter_set.hpp.
#include "filter_set.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string.h>

#include "kernels.hpp"

namespace vk {
namespace {
std::atomic<uint64_t> g_next_id{1};

// one build stream + pinned staging block per device, shared by the builds on it (they are short and serialise)
struct BuildLane {
  std::mutex mu;
  hipStream_t stream = nullptr;
  char *pin = nullptr;
  size_t pin_cap = 0;
  void *d_stage = nullptr;
  size_t d_cap = 0;
  unsigned long long *d_count = nullptr;
  unsigned long long *d_partial = nullptr;   // [kPartials]: a combine's per-block bit counts
};
constexpr size_t kPartials = 2048;
BuildLane *lane_of(int device) {
  static std::mutex mu;
  static std::vector<std::unique_ptr<BuildLane>> lanes;
  std::lock_guard<std::mutex> lk(mu);
  if ((size_t)device >= lanes.size()) lanes.resize((size_t)device + 1);
  if (!lanes[(size_t)device]) lanes[(size_t)device] = std::make_unique<BuildLane>();
  return lanes[(size_t)device].get();
}
// Bitmap blocks are recycled per device and size: a filter per FT.SEARCH (a predicate combined from cached terms) otherwise pays
// a hipMalloc per copy, and -- worse -- its release a hipFree, which waits for EVERYTHING in flight on the device: the searches
// of every other request.  Up to kPoolBytes per device wait for the next filter of their size (an index's filters are all
// of one size); beyond that a block is freed as before.
struct BitsPool {
  std::mutex mu;
  std::vector<std::pair<size_t, uint64_t *>> free;
  size_t held = 0;
};
constexpr size_t kPoolBytes = (size_t)512 << 20;
BitsPool *pool_of(int device) {
  static std::mutex mu;
  static std::vector<std::unique_ptr<BitsPool>> pools;
  std::lock_guard<std::mutex> lk(mu);
  if ((size_t)device >= pools.size()) pools.resize((size_t)device + 1);
  if (!pools[(size_t)device]) pools[(size_t)device] = std::make_unique<BitsPool>();
  return pools[(size_t)device].get();
}
// (the caller has made `device` current)
hipError_t pool_take(int device, size_t bytes, uint64_t **out) {
  BitsPool *p = pool_of(device);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    for (size_t i = p->free.size(); i-- > 0;)
      if (p->free[i].first == bytes) {
        *out = p->free[i].second;
        p->free.erase(p->free.begin() + (std::ptrdiff_t)i);
        p->held -= bytes;
        return hipSuccess;
      }
  }
  return hipMalloc(reinterpret_cast<void **>(out), bytes);
}
void pool_give(int device, size_t bytes, uint64_t *bits) {
  BitsPool *p = pool_of(device);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->held + bytes <= kPoolBytes) {
      p->free.emplace_back(bytes, bits);
      p->held += bytes;
      return;
    }
  }
  (void)hipSetDevice(device);
  (void)hipFree(bits);
}
constexpr size_t kStageBytes = (size_t)4 << 20;   // ids travel in 4 MiB pieces through pinned memory: copy k+1 is filled while k is in flight

Status lane_ready(BuildLane *l) {
  if (!l->stream) VK_HIP_TRY(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
  if (!l->pin) {
    VK_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&l->pin), 2 * kStageBytes, hipHostMallocDefault));
    l->pin_cap = 2 * kStageBytes;
  }
  if (!l->d_count) VK_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&l->d_count), 8));
  if (!l->d_partial) VK_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&l->d_partial), 2 * kPartials * 8));   // (ids | runs)
  return Status::Ok();
}
Status stage_ensure(BuildLane *l, size_t bytes) {
  if (bytes <= l->d_cap) return Status::Ok();
  if (l->d_stage) (void)hipFree(l->d_stage);
  l->d_stage = nullptr;
  l->d_cap = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
  VK_HIP_TRY(hipMalloc(&l->d_stage, want));
  l->d_cap = want;
  return Status::Ok();
}
struct LaneDrain {   // see FilterSet::build
  hipStream_t s;
  ~LaneDrain() { (void)hipStreamSynchronize(s); }
};
// host words -> device through the two halves of the pinned block
Status upload(BuildLane *l, void *d_dst, const void *h_src, size_t bytes) {
  hipEvent_t ev[2] = {nullptr, nullptr};
  VK_HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  if (hipError_t e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming); e != hipSuccess) {
    (void)hipEventDestroy(ev[0]);
    return Status::Err(4, std::string("hipEventCreate: ") + hipGetErrorString(e));
  }
  Status st = Status::Ok();
  bool used[2] = {false, false};
  size_t off = 0;
  for (int h = 0; off < bytes && st.ok(); h ^= 1) {
    const size_t n = std::min(kStageBytes, bytes - off);
    if (used[h]) {
      hipError_t e = hipEventSynchronize(ev[h]);
      if (e != hipSuccess) { st = Status::Err(4, std::string("hipEventSynchronize: ") + hipGetErrorString(e)); break; }
    }
    memcpy(l->pin + (size_t)h * kStageBytes, static_cast<const char *>(h_src) + off, n);
    hipError_t e = hipMemcpyAsync(static_cast<char *>(d_dst) + off, l->pin + (size_t)h * kStageBytes, n, hipMemcpyHostToDevice, l->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[h], l->stream);
    if (e != hipSuccess) { st = Status::Err(4, std::string("filter upload: ") + hipGetErrorString(e)); break; }
    used[h] = true;
    off += n;
  }
  (void)hipStreamSynchronize(l->stream);
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  return st;
}
}  // namespace

FilterSet::~FilterSet() {
  // (nobody reads the bitmap any more: every search that carried it held a reference until its answer was delivered)
  for (Copy &c : copies_)
    if (c.bits) pool_give(c.device, (size_t)(words() + 1) * 8, c.bits);
}

// the copies of an (uninitialised) filter of nbits bits on `devices`
Status FilterSet::allocate(const std::vector<int> &devices, uint64_t nbits, std::shared_ptr<FilterSet> *out) {
  std::shared_ptr<FilterSet> f(new FilterSet());
  f->nbits_ = nbits;
  f->id_ = g_next_id.fetch_add(1, std::memory_order_relaxed);
  const size_t alloc = ((size_t)f->words() + 1) * 8;             // (one word of slack: the kernels read whole words)
  for (int dev : devices) {
    bool have = false;
    for (const Copy &c : f->copies_) have = have || c.device == dev;
    if (have) continue;                              // (logical shards share a device)
    VK_HIP_TRY(hipSetDevice(dev));
    uint64_t *p = nullptr;
    VK_HIP_TRY(pool_take(dev, alloc, &p));
    f->copies_.push_back(Copy{dev, p});
  }
  *out = std::move(f);
  return Status::Ok();
}

Status FilterSet::build(const std::vector<int> &devices, uint64_t nbits, const uint64_t *ids, uint64_t n_ids, const uint64_t *runs,
                        uint64_t n_runs, const uint64_t *host_bits, std::shared_ptr<FilterSet> *out) {
  if (devices.empty()) return Status::Err(1, "filter: the index has no device");
  if (nbits >= ((uint64_t)1 << 40)) return Status::Err(1, "filter: nbits out of range");
  if ((n_ids && !ids) || (n_runs && !runs)) return Status::Err(1, "filter: NULL id list");
  std::shared_ptr<FilterSet> f;
  VK_TRY(allocate(devices, nbits, &f));
  const size_t words = (size_t)f->words();
  const size_t alloc = (words + 1) * 8;
  const int dev0 = f->copies_[0].device;
  BuildLane *l = lane_of(dev0);
  {
    std::lock_guard<std::mutex> lk(l->mu);
    VK_HIP_TRY(hipSetDevice(dev0));
    VK_TRY(lane_ready(l));
    // An error return below lets `f` go, and its blocks back into the pool, while copies and kernels that write them may
    // still be queued on the lane's stream: whatever way this scope is left, the stream is drained first.
    LaneDrain drained_on_exit{l->stream};
    uint64_t *bits = f->copies_[0].bits;
    VK_HIP_TRY(hipMemsetAsync(bits, 0, alloc, l->stream));
    if (host_bits && words) VK_TRY(upload(l, bits, host_bits, words * 8));
    if (host_bits && words && nbits % 64) {   // (a caller's bitmap may carry stray bits past nbits in its last word: they would be counted)
      const uint64_t last = host_bits[words - 1] & (~0ull >> (64 - nbits % 64));
      VK_HIP_TRY(hipMemcpyAsync(bits + words - 1, &last, 8, hipMemcpyHostToDevice, l->stream));
      VK_HIP_TRY(hipStreamSynchronize(l->stream));
    }
    // ids and runs side by side in the staging block; both kernels count what they turn on (a partial sum per block), a
    // base bitmap is counted by the popcount pass; one wait behind the kernels (upload() waits for its own copies)
    const size_t ids_bytes = (size_t)n_ids * 8, runs_bytes = (size_t)n_runs * 16;
    if (n_ids || n_runs) VK_TRY(stage_ensure(l, ids_bytes + runs_bytes));
    char *stage = static_cast<char *>(l->d_stage);
    uint32_t nb_ids = 0, nb_runs = 0;
    if (n_ids) {
      VK_TRY(upload(l, stage, ids, ids_bytes));
      nb_ids = filter_set_ids_blocks(n_ids);
      VK_HIP_TRY(launch_filter_set_ids(bits, nbits, reinterpret_cast<const uint64_t *>(stage), n_ids, l->d_partial, l->stream));
    }
    if (n_runs) {
      VK_TRY(upload(l, stage + ids_bytes, runs, runs_bytes));
      nb_runs = filter_set_runs_blocks(n_runs);
      VK_HIP_TRY(launch_filter_set_runs(bits, nbits, reinterpret_cast<const uint64_t *>(stage + ids_bytes), n_runs, l->d_partial + kPartials, l->stream));
    }
    if (nb_ids > kPartials || nb_runs > kPartials) return Status::Err(4, "filter: build grid larger than its partial sums");
    unsigned long long cnt = 0;
    // (the pinned block is free: upload() has waited for its copies)
    unsigned long long *h_part = reinterpret_cast<unsigned long long *>(l->pin);
    if (nb_ids) VK_HIP_TRY(hipMemcpyAsync(h_part, l->d_partial, (size_t)nb_ids * 8, hipMemcpyDeviceToHost, l->stream));
    if (nb_runs) VK_HIP_TRY(hipMemcpyAsync(h_part + kPartials, l->d_partial + kPartials, (size_t)nb_runs * 8, hipMemcpyDeviceToHost, l->stream));
    if (host_bits && words) {   // (before the id kernels in stream order would be as good: they only add bits, counted separately)
      VK_HIP_TRY(hipMemsetAsync(l->d_count, 0, 8, l->stream));
      VK_HIP_TRY(launch_filter_popcount(bits, words, l->d_count, l->stream));
      VK_HIP_TRY(hipMemcpyAsync(&cnt, l->d_count, 8, hipMemcpyDeviceToHost, l->stream));
    }
    for (size_t c = 1; c < f->copies_.size(); ++c)
      VK_HIP_TRY(hipMemcpyPeerAsync(f->copies_[c].bits, f->copies_[c].device, bits, dev0, alloc, l->stream));
    VK_HIP_TRY(hipStreamSynchronize(l->stream));
    if (!(host_bits && words)) {
      for (uint32_t i = 0; i < nb_ids; ++i) cnt += h_part[i];
      for (uint32_t i = 0; i < nb_runs; ++i) cnt += h_part[kPartials + i];
    }
    f->allowed_ = cnt;
  }
  *out = std::move(f);
  return Status::Ok();
}

Status FilterSet::combine(const FilterSet &a, const FilterSet &b, uint32_t op, std::shared_ptr<FilterSet> *out) {
  if (op > 2) return Status::Err(1, "filter: unknown combine op");
  std::vector<int> devs;
  for (const Copy &c : a.copies_)
    if (b.bits_on(c.device)) devs.push_back(c.device);
  if (devs.empty() || devs.size() != a.copies_.size() || devs.size() != b.copies_.size())
    return Status::Err(1, "filter: the two filters live on different devices");
  // both operands must cover the same label range (an allocation is zeroed up to its own slack word only, so a shorter
  // operand cannot be read beyond its end): callers build every filter of an index with one nbits
  if (a.nbits_ != b.nbits_) return Status::Err(1, "filter: combine needs filters of one size (build both with the same nbits)");
  std::shared_ptr<FilterSet> f;
  VK_TRY(allocate(devs, a.nbits_, &f));
  const size_t words = (size_t)f->words();
  // every copy's kernels are enqueued before any is waited for; the count comes back with the first copy
  unsigned long long cnt = 0;
  std::vector<BuildLane *> lanes;
  // (an error return lets `f` go while earlier copies' kernels may still be running: every lane used so far is drained)
  struct DrainAll {
    std::vector<BuildLane *> &ls; const std::vector<Copy> &cs; bool armed = true;
    ~DrainAll() {
      if (!armed) return;
      for (size_t i = 0; i < ls.size(); ++i) { (void)hipSetDevice(cs[i].device); (void)hipStreamSynchronize(ls[i]->stream); }
    }
  } drain_on_error{lanes, f->copies_};
  for (const Copy &c : f->copies_) {
    BuildLane *l = lane_of(c.device);
    std::lock_guard<std::mutex> lk(l->mu);
    VK_HIP_TRY(hipSetDevice(c.device));
    VK_TRY(lane_ready(l));
    lanes.push_back(l);
    const uint32_t nb = filter_combine_blocks(words);
    if (nb > kPartials) return Status::Err(4, "filter: combine grid larger than its partial sums");
    VK_HIP_TRY(launch_filter_combine(c.bits, a.bits_on(c.device), b.bits_on(c.device), words, op, l->d_partial, l->stream));
    if (&c == &f->copies_[0]) {
      // (the lane's pinned block: free while the lane's lock is held)
      VK_HIP_TRY(hipMemcpyAsync(l->pin, l->d_partial, (size_t)nb * 8, hipMemcpyDeviceToHost, l->stream));
      VK_HIP_TRY(hipStreamSynchronize(l->stream));
      const unsigned long long *part = reinterpret_cast<const unsigned long long *>(l->pin);
      for (uint32_t i = 0; i < nb; ++i) cnt += part[i];
    }
  }
  for (size_t i = 1; i < lanes.size(); ++i) {
    VK_HIP_TRY(hipSetDevice(f->copies_[i].device));
    VK_HIP_TRY(hipStreamSynchronize(lanes[i]->stream));
  }
  drain_on_error.armed = false;
  f->allowed_ = cnt;
  *out = std::move(f);
  return Status::Ok();
}

Status FilterSet::combine_batch(const FilterSet *const *a, const FilterSet *const *b, const uint32_t *ops, uint64_t n,
                                std::vector<std::shared_ptr<FilterSet>> *out) {
  out->clear();
  if (n == 0) return Status::Ok();
  if (n > 65535) return Status::Err(1, "filter: at most 65535 combinations per call");
  const FilterSet &f0 = *a[0];
  std::vector<int> devs;
  for (const Copy &c : f0.copies_) devs.push_back(c.device);
  for (uint64_t i = 0; i < n; ++i) {
    if (ops[i] > 2) return Status::Err(1, "filter: unknown combine op");
    if (a[i]->nbits_ != f0.nbits_ || b[i]->nbits_ != f0.nbits_)
      return Status::Err(1, "filter: combine needs filters of one size (build both with the same nbits)");
    if (a[i]->copies_.size() != devs.size() || b[i]->copies_.size() != devs.size()) return Status::Err(1, "filter: the filters live on different devices");
    for (int d : devs)
      if (!a[i]->bits_on(d) || !b[i]->bits_on(d)) return Status::Err(1, "filter: the filters live on different devices");
  }
  std::vector<std::shared_ptr<FilterSet>> res(n);
  for (uint64_t i = 0; i < n; ++i) VK_TRY(allocate(devs, f0.nbits_, &res[i]));
  const size_t words = (size_t)f0.words();
  const size_t tab_bytes = (size_t)n * 32, cnt_bytes = (size_t)n * 8;
  std::vector<BuildLane *> lanes;
  struct DrainAll {   // (an error return lets the results go while kernels that write them may still run)
    std::vector<BuildLane *> &ls; const std::vector<int> &ds; bool armed = true;
    ~DrainAll() {
      if (!armed) return;
      for (size_t i = 0; i < ls.size(); ++i) { (void)hipSetDevice(ds[i]); (void)hipStreamSynchronize(ls[i]->stream); }
    }
  } drain_on_error{lanes, devs};
  std::vector<unsigned long long> counts(n, 0);
  for (size_t di = 0; di < devs.size(); ++di) {
    BuildLane *l = lane_of(devs[di]);
    std::lock_guard<std::mutex> lk(l->mu);
    VK_HIP_TRY(hipSetDevice(devs[di]));
    VK_TRY(lane_ready(l));
    lanes.push_back(l);
    if (tab_bytes + cnt_bytes > l->pin_cap) return Status::Err(1, "filter: batch too large for the staging block");
    VK_TRY(stage_ensure(l, tab_bytes + cnt_bytes));
    uint64_t *tab = reinterpret_cast<uint64_t *>(l->pin);   // (the lane's pinned block: free while its lock is held)
    for (uint64_t i = 0; i < n; ++i) {
      tab[4 * i + 0] = reinterpret_cast<uint64_t>(res[i]->copies_[di].bits);
      tab[4 * i + 1] = reinterpret_cast<uint64_t>(a[i]->bits_on(devs[di]));
      tab[4 * i + 2] = reinterpret_cast<uint64_t>(b[i]->bits_on(devs[di]));
      tab[4 * i + 3] = ops[i];
    }
    char *stage = static_cast<char *>(l->d_stage);
    VK_HIP_TRY(hipMemcpyAsync(stage, tab, tab_bytes, hipMemcpyHostToDevice, l->stream));
    VK_HIP_TRY(hipMemsetAsync(stage + tab_bytes, 0, cnt_bytes, l->stream));
    VK_HIP_TRY(launch_filter_combine_batch(reinterpret_cast<const uint64_t *>(stage), (uint32_t)n, words,
                                           reinterpret_cast<unsigned long long *>(stage + tab_bytes), l->stream));
    // the counts come back with the first device's copy (every copy holds the same bits); every device is waited for under
    // its lane's lock: the table travels from the lane's pinned block, which the lock's next holder overwrites
    if (di == 0) VK_HIP_TRY(hipMemcpyAsync(l->pin + tab_bytes, stage + tab_bytes, cnt_bytes, hipMemcpyDeviceToHost, l->stream));
    VK_HIP_TRY(hipStreamSynchronize(l->stream));
    if (di == 0) memcpy(counts.data(), l->pin + tab_bytes, cnt_bytes);
  }
  drain_on_error.armed = false;
  for (uint64_t i = 0; i < n; ++i) res[i]->allowed_ = counts[i];
  out->swap(res);
  return Status::Ok();
}

Status FilterSet::read(uint64_t *out_words, uint64_t n_words) const {
  if (copies_.empty()) return Status::Err(4, "filter: no device copy");
  const uint64_t n = std::min<uint64_t>(n_words, words());
  VK_HIP_TRY(hipSetDevice(copies_[0].device));
  if (n) VK_HIP_TRY(hipMemcpy(out_words, copies_[0].bits, (size_t)n * 8, hipMemcpyDeviceToHost));
  for (uint64_t i = n; i < n_words; ++i) out_words[i] = 0;
  return Status::Ok();
}

// kernels.hpp: see the declaration
hipError_t ensure_max_lds(const void *fn) {
  static std::mutex mu;
  static std::vector<std::pair<const void *, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  for (const auto &d : done)
    if (d.first == fn && d.second == dev) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.emplace_back(fn, dev);
  return e;
}

}  // namespace vk
