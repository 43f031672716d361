// hip_virtual.cc -- TEST INFRASTRUCTURE: a model of the HIP runtime with N VIRTUAL DEVICES in host memory, for the code of
// csrc/sharded_index.cc / filter_set.cc that only ever runs with more than one GPU (the peer copies, one enqueue thread per
// device, events across devices, the RCCL gather) and that no box this repository was built on has: "device" memory is
// malloc'd and TAGGED with its device, a stream is a THREAD that executes what was enqueued on it in order, an event is a
// sequence number the recording stream publishes -- so a launch that is not ordered behind the copy it depends on IS a data
// race (TSAN reports it), a buffer overrun IS a heap overflow (ASAN), and a copy or a kernel that touches memory of a device
// it may not touch is recorded as a violation the test prints and fails on.  Also here: an in-process model of the five
// RCCL entry points the gather uses (barrier semantics of a collective across its ranks' streams).
//
// Linked only into tests/helpers/san_sharded_main.cc.  The model is deliberately STRICTER than the real runtime where the
// product's design allows it (a kernel may touch only memory of its own device or pinned host memory; a peer copy's stream
// must belong to one of the two devices and peer access must have been enabled).
#include "hip_virtual.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>

namespace {

constexpr int kMaxDev = 16;
std::atomic<int> g_ndev{8};
thread_local int t_dev = 0;
std::mutex g_vmu;
std::vector<std::string> g_violations;
std::atomic<uint64_t> g_peer_copies{0}, g_kernels{0}, g_collectives{0};
bool g_peer_ok[kMaxDev][kMaxDev];        // hipDeviceCanAccessPeer
bool g_peer_on[kMaxDev][kMaxDev];        // hipDeviceEnablePeerAccess done
std::once_flag g_once;
void init_tables() {
  for (int a = 0; a < kMaxDev; ++a)
    for (int b = 0; b < kMaxDev; ++b) { g_peer_ok[a][b] = a != b; g_peer_on[a][b] = false; }
}

void violation(const std::string &s) {
  std::lock_guard<std::mutex> lk(g_vmu);
  if (g_violations.size() < 64) g_violations.push_back(s);
}
[[noreturn]] void stuck(const char *what) {
  fprintf(stderr, "hip_virtual: STUCK for 30 s in %s -- an event that is never recorded, or a collective a rank never joined\n", what);
  fflush(stderr);
  abort();
}

// ---- memory ------------------------------------------------------------------------------------------------------------
struct Block { size_t bytes; int device; };   // device -1: pinned host memory
std::mutex g_mmu;
std::map<uintptr_t, Block> g_blocks;
void reg(void *p, size_t n, int dev) {
  std::lock_guard<std::mutex> lk(g_mmu);
  g_blocks[reinterpret_cast<uintptr_t>(p)] = Block{n, dev};
}
bool unreg(void *p) {
  std::lock_guard<std::mutex> lk(g_mmu);
  return g_blocks.erase(reinterpret_cast<uintptr_t>(p)) != 0;
}
// the device of [p, p + n): -1 pinned, -2 plain host memory, -3 = straddles the end of its block
int owner(const void *p, size_t n) {
  std::lock_guard<std::mutex> lk(g_mmu);
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  auto it = g_blocks.upper_bound(a);
  if (it == g_blocks.begin()) return -2;
  --it;
  if (a >= it->first + it->second.bytes) return -2;
  if (a + n > it->first + it->second.bytes) return -3;
  return it->second.device;
}

// ---- streams and events ------------------------------------------------------------------------------------------------
struct VStream {
  int device;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  uint64_t enq = 0, done = 0;
  bool stop = false;
  std::thread th;
  explicit VStream(int d) : device(d) { th = std::thread([this] { run(); }); }
  ~VStream() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    th.join();
  }
  void run() {
    for (;;) {
      std::function<void()> op;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        op = std::move(q.front());
        q.pop_front();
      }
      op();
      {
        std::lock_guard<std::mutex> lk(mu);
        ++done;
      }
      cv.notify_all();
    }
  }
  void push(std::function<void()> op) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(op)); ++enq; }
    cv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t upto = enq;
    if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return done >= upto; })) stuck("hipStreamSynchronize");
  }
  bool idle() {
    std::lock_guard<std::mutex> lk(mu);
    return done >= enq;
  }
};
struct VEvent {
  int device;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, completed = 0;
};
std::mutex g_smu;
std::set<VStream *> g_streams;
std::set<VEvent *> g_events;
VStream *S(hipStream_t s) {
  VStream *v = reinterpret_cast<VStream *>(s);
  std::lock_guard<std::mutex> lk(g_smu);
  if (!g_streams.count(v)) {
    fprintf(stderr, "hip_virtual: work enqueued on a stream that does not exist (%p; the null stream is not modelled)\n", (void *)s);
    abort();
  }
  return v;
}
VEvent *E(hipEvent_t e) {
  VEvent *v = reinterpret_cast<VEvent *>(e);
  std::lock_guard<std::mutex> lk(g_smu);
  if (!g_events.count(v)) { fprintf(stderr, "hip_virtual: unknown event %p\n", (void *)e); abort(); }
  return v;
}
std::string dev_name(int d) { return d == -1 ? "pinned host" : d == -2 ? "plain host" : d == -3 ? "OUT OF ITS BLOCK" : "device " + std::to_string(d); }

}  // namespace

// ---- hooks of the test program ---------------------------------------------------------------------------------------------
namespace hipv {
void set_device_count(int n) { std::call_once(g_once, init_tables); g_ndev = n; }
void set_peer_capable(int a, int b, bool ok) { std::call_once(g_once, init_tables); g_peer_ok[a][b] = ok; }
std::vector<std::string> take_violations() {
  std::lock_guard<std::mutex> lk(g_vmu);
  std::vector<std::string> v;
  v.swap(g_violations);
  return v;
}
uint64_t peer_copies() { return g_peer_copies.load(); }
uint64_t kernels() { return g_kernels.load(); }
uint64_t collectives() { return g_collectives.load(); }
int stream_device(hipStream_t s) { return S(s)->device; }
int memory_device(const void *p, size_t n) { return owner(p, n); }
// a "kernel": `body` runs on the stream's thread, in stream order; the launch must come from a thread whose current device
// is the stream's, and every buffer it names must lie on that device or in pinned host memory
void launch(hipStream_t s, const char *name, std::vector<Access> touched, std::function<void()> body) {
  VStream *v = S(s);
  if (t_dev != v->device)
    violation(std::string(name) + ": launched with current device " + std::to_string(t_dev) + " on a stream of device " + std::to_string(v->device));
  for (const Access &a : touched) {
    if (!a.p || !a.bytes) continue;
    const int o = owner(a.p, a.bytes);
    if (o != v->device && o != -1)
      violation(std::string(name) + " on device " + std::to_string(v->device) + ": buffer '" + a.what + "' lies in " + dev_name(o));
  }
  g_kernels.fetch_add(1);
  v->push(std::move(body));
}
}  // namespace hipv

// ---- the HIP entry points -------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipGetDeviceCount(int *n) { std::call_once(g_once, init_tables); *n = g_ndev; return hipSuccess; }
hipError_t hipSetDevice(int d) {
  std::call_once(g_once, init_tables);
  if (d < 0 || d >= g_ndev) return hipErrorInvalidDevice;
  t_dev = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_dev; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorInvalidDevice ? "invalid device ordinal" : "virtual HIP error"; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int a, int b) {
  std::call_once(g_once, init_tables);
  if (a < 0 || b < 0 || a >= g_ndev || b >= g_ndev) return hipErrorInvalidDevice;
  *can = g_peer_ok[a][b] ? 1 : 0;
  return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) {
  if (peer < 0 || peer >= g_ndev || peer == t_dev) return hipErrorInvalidDevice;
  if (!g_peer_ok[t_dev][peer]) return hipErrorInvalidDevice;
  if (g_peer_on[t_dev][peer]) return hipErrorPeerAccessAlreadyEnabled;
  g_peer_on[t_dev][peer] = true;
  return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t n) {
  *p = malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0xCD, n);   // (uninitialised device memory is not zero)
  reg(*p, n ? n : 1, t_dev);
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  // hipFree waits for the device: every stream of the block's device is drained first
  const int d = owner(p, 1);
  std::vector<VStream *> ss;
  { std::lock_guard<std::mutex> lk(g_smu); for (VStream *s : g_streams) if (s->device == d) ss.push_back(s); }
  for (VStream *s : ss) s->drain();
  if (!unreg(p)) violation("hipFree of a pointer hipMalloc did not return");
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  *p = malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0, n);
  reg(*p, n ? n : 1, -1);
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  if (!p) return hipSuccess;
  if (!unreg(p)) violation("hipHostFree of a pointer hipHostMalloc did not return");
  free(p);
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  VStream *v = new VStream(t_dev);
  { std::lock_guard<std::mutex> lk(g_smu); g_streams.insert(v); }
  *s = reinterpret_cast<hipStream_t>(v);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  VStream *v = S(s);
  v->drain();
  { std::lock_guard<std::mutex> lk(g_smu); g_streams.erase(v); }
  delete v;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->drain(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return S(s)->idle() ? hipSuccess : hipErrorNotReady; }

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
  VEvent *v = new VEvent();
  v->device = t_dev;
  { std::lock_guard<std::mutex> lk(g_smu); g_events.insert(v); }
  *e = reinterpret_cast<hipEvent_t>(v);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
  VEvent *v = E(e);
  { std::lock_guard<std::mutex> lk(g_smu); g_events.erase(v); }
  // (a stream may still hold an op that publishes this event: the real runtime defers the release likewise)
  {
    std::unique_lock<std::mutex> lk(v->mu);
    if (!v->cv.wait_for(lk, std::chrono::seconds(30), [&] { return v->completed >= v->recorded; })) stuck("hipEventDestroy");
  }
  delete v;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  VEvent *v = E(e);
  VStream *st = S(s);
  if (v->device != st->device) {   // hipErrorInvalidHandle on the real runtime
    violation("hipEventRecord: an event of device " + std::to_string(v->device) + " recorded on a stream of device " + std::to_string(st->device));
    return hipErrorInvalidHandle;
  }
  uint64_t seq;
  { std::lock_guard<std::mutex> lk(v->mu); seq = ++v->recorded; }
  st->push([v, seq] {
    { std::lock_guard<std::mutex> lk(v->mu); v->completed = std::max(v->completed, seq); }
    v->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  VEvent *v = E(e);
  uint64_t seq;
  { std::lock_guard<std::mutex> lk(v->mu); seq = v->recorded; }
  if (seq == 0) return hipSuccess;   // never recorded: no dependency
  S(s)->push([v, seq] {
    std::unique_lock<std::mutex> lk(v->mu);
    if (!v->cv.wait_for(lk, std::chrono::seconds(30), [&] { return v->completed >= seq; })) stuck("hipStreamWaitEvent");
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  VEvent *v = E(e);
  std::unique_lock<std::mutex> lk(v->mu);
  const uint64_t seq = v->recorded;
  if (!v->cv.wait_for(lk, std::chrono::seconds(30), [&] { return v->completed >= seq; })) stuck("hipEventSynchronize");
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// false: the range runs past the end of its allocation -- reported, and the copy is NOT executed
static bool check_copy_side(const char *fn, const char *side, const void *p, size_t n, int want_dev, bool host_side) {
  const int o = owner(p, n);
  if (o == -3) { violation(std::string(fn) + ": " + side + " runs past the end of its allocation"); return false; }
  if (host_side ? (o >= 0) : (o != want_dev))
    violation(std::string(fn) + ": " + side + " lies in " + dev_name(o) + ", expected " + (host_side ? std::string("host memory") : dev_name(want_dev)));
  return true;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind kind, hipStream_t st) {
  VStream *v = S(st);
  if (n == 0) return hipSuccess;
  bool ok = true;
  if (kind == hipMemcpyHostToDevice) { ok &= check_copy_side("hipMemcpyAsync(H2D)", "dst", d, n, v->device, false); ok &= check_copy_side("hipMemcpyAsync(H2D)", "src", s, n, 0, true); }
  else if (kind == hipMemcpyDeviceToHost) { ok &= check_copy_side("hipMemcpyAsync(D2H)", "src", s, n, v->device, false); ok &= check_copy_side("hipMemcpyAsync(D2H)", "dst", d, n, 0, true); }
  else if (kind == hipMemcpyDeviceToDevice) { ok &= check_copy_side("hipMemcpyAsync(D2D)", "dst", d, n, v->device, false); ok &= check_copy_side("hipMemcpyAsync(D2D)", "src", s, n, v->device, false); }
  if (!ok) return hipErrorInvalidValue;
  // A pageable host source is copied by the runtime before the call returns; a pinned one is read when the stream gets there
  if (kind == hipMemcpyHostToDevice && owner(s, n) == -2) {
    std::vector<char> staged(static_cast<const char *>(s), static_cast<const char *>(s) + n);
    v->push([d, staged = std::move(staged)] { memcpy(d, staged.data(), staged.size()); });
  } else {
    v->push([d, s, n] { memmove(d, s, n); });
  }
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int val, size_t n, hipStream_t st) {
  VStream *v = S(st);
  if (n == 0) return hipSuccess;
  if (!check_copy_side("hipMemsetAsync", "dst", d, n, v->device, false)) return hipErrorInvalidValue;
  v->push([d, val, n] { memset(d, val, n); });
  return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *d, int ddev, const void *s, int sdev, size_t n, hipStream_t st) {
  VStream *v = S(st);
  if (n == 0) return hipSuccess;
  bool ok = check_copy_side("hipMemcpyPeerAsync", "dst", d, n, ddev, false);
  ok &= check_copy_side("hipMemcpyPeerAsync", "src", s, n, sdev, false);
  if (!ok) return hipErrorInvalidValue;
  if (v->device != ddev && v->device != sdev)
    violation("hipMemcpyPeerAsync " + std::to_string(sdev) + " -> " + std::to_string(ddev) + " on a stream of device " + std::to_string(v->device));
  const int other = v->device == ddev ? sdev : ddev;
  if (other != v->device && !g_peer_on[v->device][other])
    violation("hipMemcpyPeerAsync " + std::to_string(sdev) + " -> " + std::to_string(ddev) + ": peer access " + std::to_string(v->device) + " -> " +
              std::to_string(other) + " was never enabled (the copy would be staged through host memory)");
  if (ddev != sdev) g_peer_copies.fetch_add(1);
  v->push([d, s, n] { memmove(d, s, n); });
  return hipSuccess;
}
}  // extern "C"

// ---- RCCL model -----------------------------------------------------------------------------------------------------------
namespace {
struct Clique;
struct Comm { Clique *clique; int rank, nranks, device; };
struct Clique { std::vector<Comm *> comms; };
struct Collective {   // one all-gather across the ranks of a clique
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0, finished = 0;
  std::vector<const void *> send;
  std::vector<void *> recv;
  size_t bytes = 0;   // per rank
};
struct PendingOp { Comm *c; const void *send; void *recv; size_t bytes; hipStream_t stream; };
thread_local int t_group_depth = 0;
thread_local std::vector<PendingOp> t_group_ops;

int flush_group() {
  // the i-th all-gather of every rank forms one collective (ranks issue the same sequence)
  std::map<Clique *, std::vector<std::vector<PendingOp>>> by_clique;   // clique -> rank -> ops in order
  for (const PendingOp &op : t_group_ops) {
    auto &v = by_clique[op.c->clique];
    if (v.empty()) v.resize(op.c->nranks);
    v[op.c->rank].push_back(op);
  }
  t_group_ops.clear();
  for (auto &kv : by_clique) {
    const auto &per_rank = kv.second;
    const size_t n_ops = per_rank[0].size();
    for (const auto &r : per_rank)
      if (r.size() != n_ops) { violation("RCCL group: the ranks of one communicator set issued different numbers of collectives (a rank is missing: the real call hangs)"); return 1; }
    for (size_t i = 0; i < n_ops; ++i) {
      auto col = std::make_shared<Collective>();
      col->n = (int)per_rank.size();
      col->bytes = per_rank[0][i].bytes;
      for (const auto &r : per_rank) {
        if (r[i].bytes != col->bytes) { violation("ncclAllGather: ranks disagree on the count"); return 1; }
        col->send.push_back(r[i].send);
        col->recv.push_back(r[i].recv);
      }
      g_collectives.fetch_add(1);
      for (int rk = 0; rk < col->n; ++rk) {
        reinterpret_cast<VStream *>(per_rank[rk][i].stream)->push([col, rk] {
          std::unique_lock<std::mutex> lk(col->mu);
          ++col->arrived;   // this rank's stream got here: its send buffer is ready
          col->cv.notify_all();
          if (!col->cv.wait_for(lk, std::chrono::seconds(30), [&] { return col->arrived == col->n; })) stuck("ncclAllGather (a rank's stream never reached the collective)");
          lk.unlock();
          for (int src = 0; src < col->n; ++src) memcpy(static_cast<char *>(col->recv[rk]) + (size_t)src * col->bytes, col->send[src], col->bytes);
          lk.lock();
          ++col->finished;  // nobody's send buffer is reused before every rank has read it
          col->cv.notify_all();
          if (!col->cv.wait_for(lk, std::chrono::seconds(30), [&] { return col->finished == col->n; })) stuck("ncclAllGather (completion)");
        });
      }
    }
  }
  return 0;
}
}  // namespace

// (the prototypes are the real header's where it is installed: the product calls these through pointers of exactly those types)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclUint64 = 5, ncclFloat32 = 7 } ncclDataType_t;
#endif
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
  std::set<int> seen;
  for (int i = 0; i < n; ++i)
    if (devs[i] < 0 || devs[i] >= g_ndev || !seen.insert(devs[i]).second) { violation("ncclCommInitAll: device list with a duplicate or an unknown device"); return ncclInvalidArgument; }
  Clique *q = new Clique();
  for (int i = 0; i < n; ++i) {
    Comm *c = new Comm{q, i, n, devs[i]};
    q->comms.push_back(c);
    comms[i] = reinterpret_cast<ncclComm_t>(c);
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  Comm *cm = reinterpret_cast<Comm *>(c);
  Clique *q = cm->clique;
  q->comms[cm->rank] = nullptr;
  delete cm;
  if (std::all_of(q->comms.begin(), q->comms.end(), [](Comm *x) { return x == nullptr; })) delete q;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart(void) { ++t_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
  if (--t_group_depth > 0) return ncclSuccess;
  return flush_group() ? ncclInvalidArgument : ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dtype, ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  const size_t es = dtype == ncclFloat32 ? 4 : dtype == ncclUint64 ? 8 : 0;
  if (!es) { violation("ncclAllGather: a data type the gather does not use"); return ncclInvalidArgument; }
  VStream *v = S(stream);
  if (v->device != c->device) violation("ncclAllGather: rank " + std::to_string(c->rank) + " (device " + std::to_string(c->device) + ") on a stream of device " + std::to_string(v->device));
  if (owner(send, count * es) != c->device) violation("ncclAllGather: send buffer of rank " + std::to_string(c->rank) + " lies in " + dev_name(owner(send, count * es)));
  if (owner(recv, count * es * c->nranks) != c->device) violation("ncclAllGather: receive buffer of rank " + std::to_string(c->rank) + " lies in " + dev_name(owner(recv, count * es * c->nranks)) + " or is too small for nranks x count");
  t_group_ops.push_back(PendingOp{c, send, recv, count * es, stream});
  if (t_group_depth == 0) return flush_group() ? ncclInvalidArgument : ncclSuccess;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t) { return "virtual RCCL error"; }
// the seam of csrc/sharded_index.cc: load the gather's entry points from this program
const char *vk_test_rccl_library() { return nullptr; }
}
