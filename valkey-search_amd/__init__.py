"""valkey-search_amd: MI355X-native vector-kNN backend for valkey-search.

The product is libvkindex.so (HIP kernels + the C ABI of include/vk_index.h) and the C++
host classes in csrc/host/.  This Python module is only the ctypes binding the tests and
bench.py drive it with -- it contains no search logic and NO fallback: if the shared
library is missing, or there is no gfx950 device, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
# VKINDEX_LIB: load another build of the same ABI (A/B timing of kernel variants on one GPU box)
LIB_PATH = Path(os.environ["VKINDEX_LIB"]) if os.environ.get("VKINDEX_LIB") else PKG_DIR / "libvkindex.so"
HOST_LIB_PATH = PKG_DIR / "libvkhost.so"

VK_OK, VK_ERR_INVALID, VK_ERR_CAPACITY, VK_ERR_NOT_FOUND, VK_ERR_INTERNAL, VK_ERR_CANCELLED, VK_ERR_NO_DEVICE, VK_ERR_BUSY = range(8)
ALGO = {"FLAT": 0, "HNSW": 1}
METRIC = {"L2": 0, "IP": 1, "COSINE": 2}


class VkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vk_status {code}: {msg}")
        self.code = code
        self.msg = msg


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("algo", C.c_uint32), ("metric", C.c_uint32), ("dtype", C.c_uint32),
                ("dim", C.c_uint32), ("block_size", C.c_uint32), ("initial_cap", C.c_uint64), ("m", C.c_uint32),
                ("ef_construction", C.c_uint32), ("ef_runtime", C.c_uint32), ("allow_replace_deleted", C.c_uint32),
                ("random_seed", C.c_uint64), ("device_id", C.c_int32), ("build_threads", C.c_uint32),
                ("n_shards", C.c_uint32), ("shard_devices", C.c_int32 * 16), ("shard_ef_pct", C.c_uint32), ("load_skip_validation", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("count", C.c_uint64), ("deleted", C.c_uint64), ("capacity", C.c_uint64),
                ("device_bytes", C.c_uint64), ("host_bytes", C.c_uint64), ("staged_ops", C.c_uint64),
                ("max_level", C.c_int32), ("entry_point", C.c_uint32), ("last_n_eval", C.c_uint64),
                ("last_n_hops", C.c_uint64), ("last_frontier_redo", C.c_uint64), ("last_frontier_dropped", C.c_uint64),
                ("last_filter_candidates", C.c_uint64), ("last_filter_fallback", C.c_uint64),
                ("filter_batches", C.c_uint64), ("filter_kernel_ns", C.c_uint64),
                ("coalesced_batches", C.c_uint64), ("coalesced_queries", C.c_uint64),
                ("fanout_calls", C.c_uint64), ("fanout_enqueue_ns", C.c_uint64),
                ("searches", C.c_uint64), ("search_calls", C.c_uint64), ("search_errors", C.c_uint64 * 8),
                ("total_n_eval", C.c_uint64), ("total_n_hops", C.c_uint64), ("tombstoned_bytes", C.c_uint64),
                ("latency_hist", C.c_uint64 * 16), ("latency_sum_ns", C.c_uint64),
                ("submitted", C.c_uint64), ("rejected", C.c_uint64), ("queued_now", C.c_uint64),
                ("max_batches_in_flight", C.c_uint64), ("rccl_gathers", C.c_uint64),
                ("last_filter_reranked", C.c_uint64), ("max_label", C.c_uint64), ("cancelled_early", C.c_uint64),
                ("filters_built", C.c_uint64), ("filter_cache_hits", C.c_uint64), ("filter_cache_misses", C.c_uint64),
                ("filter_cache_entries", C.c_uint64), ("filter_cache_bytes", C.c_uint64),
                ("staged_adds", C.c_uint64), ("staged_adds_device", C.c_uint64), ("last_visited_mode", C.c_uint64),
                ("dispatch_idle_us", C.c_uint64), ("dispatch_window_us", C.c_uint64), ("dispatch_search_us", C.c_uint64),
                ("dispatch_handout_us", C.c_uint64), ("dispatch_completer_us", C.c_uint64),
                ("last_filter_final_rows", C.c_uint64)]


WRITE_CHUNK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
READ_CHUNK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))
SEARCH_DONE = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
ROW_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)


EXP_LIB_PATH = PKG_DIR / "libvkindex_exp.so"


def build_experiments(verbose: bool = False) -> Path:
    """The -DVK_EXPERIMENTS build of the library (ablation / cycle-counter kernels whose answers are invalid, A/B
    constants from the environment): scripts/ only -- select it with VKINDEX_LIB=<this path> in a fresh process."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-s", "-j8", "-C", str(PKG_DIR / "csrc"), "experiments"], stdout=out)
    return EXP_LIB_PATH


def build(verbose: bool = False) -> None:
    """Compile every HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-s", "-j8", "-C", str(PKG_DIR / "csrc")], stdout=out)
    if (PKG_DIR / "csrc" / "host" / "Makefile").exists():
        subprocess.check_call(["make", "-s", "-j8", "-C", str(PKG_DIR / "csrc" / "host")], stdout=out)


_lib = None


def lib() -> C.CDLL:
    """The C ABI.  Raises if libvkindex.so has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(str(LIB_PATH))
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    u64p, f32p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.vk_last_error.restype = C.c_char_p
    L.vk_abi_struct_size.argtypes = [i32]
    L.vk_abi_struct_size.restype = u64
    if L.vk_abi_struct_size(0) != C.sizeof(Params) or L.vk_abi_struct_size(1) != C.sizeof(Stats):
        raise RuntimeError(f"{LIB_PATH}: built from another vk_index.h than this binding (params {L.vk_abi_struct_size(0)} / {C.sizeof(Params)}, "
                           f"stats {L.vk_abi_struct_size(1)} / {C.sizeof(Stats)} bytes): run `python -c 'import __graft_entry__ as g; g.build()'`")
    L.vk_device_count.restype = i32
    L.vk_index_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.vk_index_destroy.argtypes = [vp]
    L.vk_index_destroy.restype = None
    L.vk_index_add.argtypes = [vp, u64, vp]
    L.vk_index_add_batch.argtypes = [vp, vp, vp, u64]
    L.vk_index_remove.argtypes = [vp, u64]
    L.vk_index_resize.argtypes = [vp, u64]
    L.vk_index_set_ef.argtypes = [vp, u32]
    L.vk_index_flush.argtypes = [vp]
    L.vk_index_set_coalescing.argtypes = [vp, u32, u32]
    L.vk_index_set_option.argtypes = [vp, C.c_char_p, u64]
    L.vk_index_get_option.argtypes = [vp, C.c_char_p, u64p]
    L.vk_index_search_submit.argtypes = [vp, vp, u64, u64, vp, u64, vp, i32, vp, vp, vp, SEARCH_DONE, vp]
    L.vk_index_shard_stats.argtypes = [vp, u32, C.POINTER(Stats)]
    L.vk_index_search.argtypes = [vp, vp, u64, u64, vp, u64, vp, i32, vp, vp, u64p]
    L.vk_index_search_batch.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, i32, vp, vp, vp]
    L.vk_index_search_batch_filters.argtypes = [vp, vp, u64, u64, u64, vp, vp, vp, i32, vp, vp, vp]
    L.vk_index_search_batch_device.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, vp, vp, vp]
    L.vk_index_search_labels.argtypes = [vp, vp, u64, vp, u64, vp, vp, u64p]
    L.vk_index_distance.argtypes = [vp, u64, vp, f32p]
    L.vk_index_get_row.argtypes = [vp, u64, vp]
    L.vk_index_contains.argtypes = [vp, u64, C.POINTER(i32)]
    L.vk_index_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.vk_index_device_rows.argtypes = [vp, u64, C.POINTER(vp), u64p]
    L.vk_index_commit_device_rows.argtypes = [vp, u64, vp]
    L.vk_merge_topk_device.argtypes = [vp, vp, u32, u64, u64, vp, vp, vp, i32, vp]
    L.vk_index_shard_count.argtypes = [vp, u32p]
    L.vk_index_shard_device_rows.argtypes = [vp, u32, u64, C.POINTER(vp), u64p]
    L.vk_index_shard_commit_device_rows.argtypes = [vp, u32, u64, vp]
    L.vk_index_save.argtypes = [vp, WRITE_CHUNK, vp]
    L.vk_index_load.argtypes = [C.POINTER(Params), READ_CHUNK, vp, C.POINTER(vp)]
    L.vk_index_load_tracked.argtypes = [C.POINTER(Params), READ_CHUNK, vp, ROW_FN, vp, C.POINTER(vp)]
    L.vk_filter_create.argtypes = [vp, u64, vp, u64, vp, u64, vp, C.POINTER(vp)]
    L.vk_filter_combine.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
    L.vk_filter_combine_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), u64, C.POINTER(vp)]
    L.vk_filter_retain.argtypes = [vp]
    L.vk_filter_retain.restype = None
    L.vk_filter_release.argtypes = [vp]
    L.vk_filter_release.restype = None
    L.vk_filter_info.argtypes = [vp, u64p, u64p]
    L.vk_filter_read.argtypes = [vp, vp, u64]
    L.vk_index_filter_cache_get.argtypes = [vp, C.c_char_p, u64, u64, C.POINTER(vp)]
    L.vk_index_filter_cache_put.argtypes = [vp, C.c_char_p, u64, u64, vp]
    L.vk_index_search_filter.argtypes = [vp, vp, u64, u64, vp, vp, i32, vp, vp, u64p]
    L.vk_index_search_submit_filter.argtypes = [vp, vp, u64, u64, vp, vp, i32, vp, vp, vp, SEARCH_DONE, vp]
    L.vk_index_search_batch_filter_handles.argtypes = [vp, vp, u64, u64, u64, vp, vp, i32, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc != VK_OK:
        raise VkError(rc, lib().vk_last_error().decode(errors="replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


DTYPE = {"f32": 0, "bf16": 1}


def make_params(algo, dim, metric, initial_cap, block_size=1024, m=16, ef_construction=200, ef_runtime=10,
                seed=100, allow_replace_deleted=False, device_id=-1, build_threads=0, dtype="f32", shard_devices=None,
                shard_ef_pct=0, load_skip_validation=False) -> Params:
    """shard_devices: list of HIP device ordinals, one sub-index per entry (a device may repeat: logical shards)"""
    p = Params(C.sizeof(Params), ALGO[algo], METRIC[metric], DTYPE[dtype], dim, block_size, initial_cap, m, ef_construction,
               ef_runtime, int(allow_replace_deleted), seed, device_id, build_threads)
    p.shard_ef_pct = int(shard_ef_pct)
    p.load_skip_validation = int(bool(load_skip_validation))
    if shard_devices:
        p.n_shards = len(shard_devices)
        for i, d in enumerate(shard_devices):
            p.shard_devices[i] = int(d)
    return p


class _Pending:
    """buffers of one submitted request"""

    def __init__(self, q, k, allow, cancel):
        self.q, self.allow, self.cancel = q, allow, cancel
        self.d = np.empty(k, np.float32)
        self.l = np.empty(k, np.uint64)
        self.n = np.zeros(1, np.uint64)
        self.status = None
        self.cb = None

    def result(self):
        m = int(self.n[0])
        return self.d[:m], self.l[:m]


_PENDING = set()


class Filter:
    """A device-resident allow-set (vk_filter_*): built on the device from id lists / id runs, reference counted."""

    def __init__(self, handle):
        self._h = handle

    def release(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.vk_filter_release(h)

    __del__ = release

    def info(self):
        nb, al = C.c_uint64(), C.c_uint64()
        _check(lib().vk_filter_info(self._h, C.byref(nb), C.byref(al)))
        return nb.value, al.value

    def read(self, n_words=None):
        nb, _ = self.info()
        w = (nb + 63) // 64 if n_words is None else n_words
        out = np.zeros(max(w, 1), np.uint64)
        _check(lib().vk_filter_read(self._h, _ptr(out), w))
        return out[:w]


class Index:
    """Thin RAII wrapper over vk_index_* (one per hnswlib algorithm object)."""

    def __init__(self, algo, dim, metric="L2", initial_cap=1024, options=None, **kw):
        """options: {name: value} applied with vk_index_set_option right after creation (csrc/options.hpp)"""
        self.algo, self.dim, self.metric = algo, dim, metric
        self.params = make_params(algo, dim, metric, initial_cap, **kw)
        h = C.c_void_p()
        _check(lib().vk_index_create(C.byref(self.params), C.byref(h)))
        self._h = h
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        _check(lib().vk_index_set_option(self._h, name.replace("_", "-").encode(), int(value)))

    def get_option(self, name) -> int:
        v = C.c_uint64()
        _check(lib().vk_index_get_option(self._h, name.replace("_", "-").encode(), C.byref(v)))
        return v.value

    def shard_stats(self, shard) -> "Stats":
        s = Stats()
        _check(lib().vk_index_shard_stats(self._h, int(shard), C.byref(s)))
        return s

    def submit(self, q, k, done, ef=0, allow=None, allow_nbits=None, cancel=None, partial_ok=True):
        """vk_index_search_submit.  Returns a handle whose arrays (.d, .l, .n) hold the answer once `done(status)` has been
        called (from a library thread); the handle keeps every buffer of the request alive."""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        h = _Pending(q, k, allow, cancel)
        ap, nb = (None, 0) if allow is None else (allow.ctypes.data, int(allow_nbits if allow_nbits is not None else allow.size * 64))
        cflag = None if cancel is None else C.cast(C.pointer(cancel), C.c_void_p)

        def _cb(_user, status, h=h, done=done):
            h.status = status
            done(status)
            _PENDING.discard(h)

        h.cb = SEARCH_DONE(_cb)
        _PENDING.add(h)
        rc = lib().vk_index_search_submit(self._h, q.ctypes.data, int(k), int(ef), ap, nb, cflag, int(partial_ok), h.d.ctypes.data,
                                          h.l.ctypes.data, h.n.ctypes.data, h.cb, None)
        if rc != VK_OK:
            _PENDING.discard(h)
            _check(rc)
        return h

    # ---- device-resident filters
    def make_filter(self, nbits, labels=None, runs=None, base_bits=None) -> Filter:
        """vk_filter_create: labels = ids in any order (duplicates fine), runs = [[first, last], ...] inclusive"""
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
        rn = None if runs is None else np.ascontiguousarray(runs, dtype=np.uint64).reshape(-1, 2)
        bb = None if base_bits is None else np.ascontiguousarray(base_bits, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().vk_filter_create(self._h, int(nbits), _ptr(lab), 0 if lab is None else lab.size, _ptr(rn),
                                      0 if rn is None else rn.shape[0], _ptr(bb), C.byref(h)))
        return Filter(h)

    def combine_filters(self, a: Filter, b: Filter, op) -> Filter:
        h = C.c_void_p()
        _check(lib().vk_filter_combine(self._h, a._h, b._h, {"and": 0, "or": 1, "andnot": 2}[op], C.byref(h)))
        return Filter(h)

    def combine_filters_batch(self, pairs, op="or"):
        """vk_filter_combine_batch: [(a, b), ...] (or (a, b, op)) -> one Filter each, one launch and one wait per device"""
        n = len(pairs)
        code = {"and": 0, "or": 1, "andnot": 2}
        A = (C.c_void_p * n)(*[p[0]._h for p in pairs])
        B = (C.c_void_p * n)(*[p[1]._h for p in pairs])
        O = (C.c_uint32 * n)(*[code[p[2] if len(p) > 2 else op] for p in pairs])
        out = (C.c_void_p * n)()
        _check(lib().vk_filter_combine_batch(self._h, A, B, O, n, out))
        return [Filter(C.c_void_p(out[i])) for i in range(n)]

    def filter_cache_get(self, key: bytes, epoch: int):
        h = C.c_void_p()
        _check(lib().vk_index_filter_cache_get(self._h, key, len(key), int(epoch), C.byref(h)))
        return Filter(h) if h.value else None

    def filter_cache_put(self, key: bytes, epoch: int, f: Filter):
        _check(lib().vk_index_filter_cache_put(self._h, key, len(key), int(epoch), f._h))

    def search_filter(self, q, k, f, ef=0, cancel=None, partial_ok=True):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        d = np.empty(max(k, 1), np.float32)
        l = np.empty(max(k, 1), np.uint64)
        n = C.c_uint64(0)
        cflag = None if cancel is None else C.cast(C.pointer(cancel), C.c_void_p)
        _check(lib().vk_index_search_filter(self._h, q.ctypes.data, int(k), int(ef), None if f is None else f._h, cflag, int(partial_ok),
                                            d.ctypes.data, l.ctypes.data, C.byref(n)))
        return d[:n.value], l[:n.value]

    def search_batch_filter_handles(self, Q, k, filters, ef=0):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = Q.shape[0]
        tab = (C.c_void_p * nq)(*[None if f is None else f._h for f in filters])
        od = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        ol = np.full((nq, max(k, 1)), np.iinfo(np.uint64).max, dtype=np.uint64)
        on = np.zeros(nq, dtype=np.uint64)
        _check(lib().vk_index_search_batch_filter_handles(self._h, _ptr(Q), nq, k, ef, tab, None, 1, _ptr(od), _ptr(ol), _ptr(on)))
        return od[:, :k], ol[:, :k], on

    def submit_filter(self, q, k, done, f, ef=0, cancel=None, partial_ok=True):
        """vk_index_search_submit_filter (the request keeps the filter alive)"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        h = _Pending(q, k, f, cancel)
        cflag = None if cancel is None else C.cast(C.pointer(cancel), C.c_void_p)

        def _cb(_user, status, h=h, done=done):
            h.status = status
            done(status)
            _PENDING.discard(h)

        h.cb = SEARCH_DONE(_cb)
        _PENDING.add(h)
        rc = lib().vk_index_search_submit_filter(self._h, q.ctypes.data, int(k), int(ef), None if f is None else f._h, cflag, int(partial_ok),
                                                 h.d.ctypes.data, h.l.ctypes.data, h.n.ctypes.data, h.cb, None)
        if rc != VK_OK:
            _PENDING.discard(h)
            _check(rc)
        return h

    @classmethod
    def _from_handle(cls, h, params, algo, dim, metric):
        self = cls.__new__(cls)
        self._h, self.params, self.algo, self.dim, self.metric = h, params, algo, dim, metric
        return self

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:          # at interpreter shutdown the module globals may be gone
            try:
                _lib.vk_index_destroy(h)
            except Exception:
                pass

    __del__ = close

    # ---- mutations
    def add(self, label, row):
        row = np.ascontiguousarray(row, dtype=np.float32)
        assert row.size == self.dim
        return lib().vk_index_add(self._h, int(label), _ptr(row))

    def add_batch(self, rows, labels=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.dim
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
        _check(lib().vk_index_add_batch(self._h, _ptr(lab), _ptr(rows), rows.shape[0]))

    def remove(self, label):
        return lib().vk_index_remove(self._h, int(label))

    def resize(self, n):
        _check(lib().vk_index_resize(self._h, int(n)))

    def set_ef(self, ef):
        _check(lib().vk_index_set_ef(self._h, int(ef)))

    def flush(self):
        _check(lib().vk_index_flush(self._h))

    def set_coalescing(self, max_batch, max_wait_us):
        """Merge concurrent single-query searches into device batches (vk_index_set_coalescing)."""
        _check(lib().vk_index_set_coalescing(self._h, int(max_batch), int(max_wait_us)))

    def search_one(self, q, k, ef=0, allow=None, allow_nbits=None, cancel=None, partial_ok=True):
        """vk_index_search itself (the per-FT.SEARCH entry point; ctypes drops the GIL around it)."""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        d = np.empty(k, np.float32)
        l = np.empty(k, np.uint64)
        n = C.c_uint64(0)
        ap, nb = (None, 0) if allow is None else (allow.ctypes.data, int(allow_nbits if allow_nbits is not None else allow.size * 64))
        cflag = None if cancel is None else C.cast(C.pointer(cancel), C.c_void_p)
        _check(lib().vk_index_search(self._h, q.ctypes.data, int(k), int(ef), ap, nb, cflag, int(partial_ok), d.ctypes.data,
                                     l.ctypes.data, C.byref(n)))
        return d[:n.value], l[:n.value]

    def search_batch_filters(self, Q, k, allows, nbits, ef=0):
        """one allow-bitmap (uint64 array or None) per query: vk_index_search_batch_filters"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = Q.shape[0]
        keep = [None if a is None else np.ascontiguousarray(a, dtype=np.uint64) for a in allows]
        tab = (C.c_void_p * nq)(*[None if a is None else a.ctypes.data for a in keep])
        nb = np.array([0 if a is None else int(b) for a, b in zip(keep, nbits)], dtype=np.uint64)
        od = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        ol = np.full((nq, max(k, 1)), np.iinfo(np.uint64).max, dtype=np.uint64)
        on = np.zeros(nq, dtype=np.uint64)
        _check(lib().vk_index_search_batch_filters(self._h, _ptr(Q), nq, k, ef, tab, _ptr(nb), None, 1, _ptr(od), _ptr(ol), _ptr(on)))
        return od[:, :k], ol[:, :k], on

    # ---- queries
    def search(self, q, k, ef=0, allow=None, allow_nbits=None, cancel=None, partial_ok=True):
        d, l, n = self.search_batch(np.asarray(q, dtype=np.float32).reshape(1, -1), k, ef, allow, allow_nbits,
                                    cancel, partial_ok)
        return d[0, :n[0]], l[0, :n[0]]

    def search_batch(self, Q, k, ef=0, allow=None, allow_nbits=None, cancel=None, partial_ok=True):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = Q.shape[0]
        od = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        ol = np.full((nq, max(k, 1)), np.iinfo(np.uint64).max, dtype=np.uint64)
        on = np.zeros(nq, dtype=np.uint64)
        if allow is not None:
            allow = np.ascontiguousarray(allow, dtype=np.uint64)
            nbits = int(allow_nbits if allow_nbits is not None else allow.size * 64)
        else:
            nbits = 0
        cflag = None if cancel is None else C.cast(C.pointer(cancel), C.c_void_p)
        _check(lib().vk_index_search_batch(self._h, _ptr(Q), nq, k, ef, _ptr(allow), nbits, cflag, int(partial_ok),
                                           _ptr(od), _ptr(ol), _ptr(on)))
        return od[:, :k], ol[:, :k], on

    def search_labels(self, q, k, labels):
        q = np.ascontiguousarray(q, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.uint64)
        od = np.empty(max(k, 1), np.float32)
        ol = np.empty(max(k, 1), np.uint64)
        n = C.c_uint64()
        _check(lib().vk_index_search_labels(self._h, _ptr(q), k, _ptr(labels), labels.size, _ptr(od), _ptr(ol),
                                            C.byref(n)))
        return od[:n.value].copy(), ol[:n.value].copy()

    def distance(self, label, q):
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = C.c_float()
        rc = lib().vk_index_distance(self._h, int(label), _ptr(q), C.byref(out))
        return None if rc == VK_ERR_NOT_FOUND else (_check(rc) or np.float32(out.value))

    def get_row(self, label):
        out = np.empty(self.dim, np.float32)
        rc = lib().vk_index_get_row(self._h, int(label), _ptr(out))
        return None if rc == VK_ERR_NOT_FOUND else (_check(rc) or out)

    def contains(self, label) -> bool:
        f = C.c_int()
        _check(lib().vk_index_contains(self._h, int(label), C.byref(f)))
        return bool(f.value)

    def stats(self) -> Stats:
        s = Stats()
        _check(lib().vk_index_get_stats(self._h, C.byref(s)))
        return s

    # ---- device-resident paths (pointers are raw device addresses, e.g. torch .data_ptr())
    def device_rows(self, n):
        p, stride = C.c_void_p(), C.c_uint64()
        _check(lib().vk_index_device_rows(self._h, n, C.byref(p), C.byref(stride)))
        return p.value, stride.value

    def commit_device_rows(self, n, labels=None):
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
        _check(lib().vk_index_commit_device_rows(self._h, n, _ptr(lab)))

    def shard_count(self) -> int:
        n = C.c_uint32()
        _check(lib().vk_index_shard_count(self._h, C.byref(n)))
        return n.value

    def shard_device_rows(self, shard, n):
        p, stride = C.c_void_p(), C.c_uint64()
        _check(lib().vk_index_shard_device_rows(self._h, shard, n, C.byref(p), C.byref(stride)))
        return p.value, stride.value

    def shard_commit_device_rows(self, shard, n, labels):
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        _check(lib().vk_index_shard_commit_device_rows(self._h, shard, n, _ptr(lab)))

    def search_batch_device(self, d_queries, nq, k, d_out_dist, d_out_label, d_out_n, ef=0, d_allow=None,
                            allow_nbits=0, stream=None):
        _check(lib().vk_index_search_batch_device(self._h, d_queries, nq, k, ef, d_allow, allow_nbits, d_out_dist,
                                                  d_out_label, d_out_n, stream))

    # ---- persistence
    def save_raw(self, callback, user):
        """vk_index_save with a C callback address (e.g. the test oracle's chunk sink): returns the status code."""
        fn = C.cast(callback, WRITE_CHUNK)
        return lib().vk_index_save(self._h, fn, user)

    def save(self):
        chunks = []

        @WRITE_CHUNK
        def wr(_u, data, n):
            chunks.append(C.string_at(data, n))
            return 0

        _check(lib().vk_index_save(self._h, wr, None))
        return chunks

    @classmethod
    def load(cls, chunks, algo, dim, metric="L2", **kw):
        kw_on_row = kw.pop("on_row", None)   # vk_index_load_tracked: on_row(label, row) per loaded element
        params = make_params(algo, dim, metric, kw.pop("initial_cap", 0), **kw)
        it = iter(chunks)

        @READ_CHUNK
        def rd(_u, buf, cap, out_len):
            try:
                c = next(it)
            except StopIteration:
                return 1
            if len(c) > cap:
                return 2
            C.memmove(buf, c, len(c))
            out_len[0] = len(c)
            return 0

        h = C.c_void_p()
        on_row = kw_on_row
        if on_row is None:
            _check(lib().vk_index_load(C.byref(params), rd, None, C.byref(h)))
        else:
            @ROW_FN
            def row_cb(_u, label, row):
                return int(on_row(int(label), np.frombuffer(C.string_at(row, dim * 4), dtype=np.float32)) or 0)

            _check(lib().vk_index_load_tracked(C.byref(params), rd, None, row_cb, None, C.byref(h)))
        return cls._from_handle(h, params, algo, dim, metric)


def merge_topk_device(d_dist, d_label, parts, nq, k, d_out_dist, d_out_label, d_out_n, device_id=-1, stream=None):
    _check(lib().vk_merge_topk_device(d_dist, d_label, parts, nq, k, d_out_dist, d_out_label, d_out_n, device_id,
                                      stream))


# ---- scripts/serving_probe.cc: native drivers of the single-query serving path against an index this process holds ----
class ProbeResult(C.Structure):
    _fields_ = [("qps", C.c_double), ("seconds", C.c_double), ("completed", C.c_uint64), ("rejected", C.c_uint64),
                ("mismatches", C.c_uint64), ("errors", C.c_uint64), ("device_batches", C.c_uint64),
                ("max_batches_in_flight", C.c_uint64), ("mean_batch", C.c_double), ("p50_us", C.c_double),
                ("p99_us", C.c_double), ("max_us", C.c_double)]

    def as_dict(self):
        return {k: (round(getattr(self, k), 1) if isinstance(getattr(self, k), float) else int(getattr(self, k))) for k, _ in self._fields_}


_probe = None


def probe_lib():
    global _probe
    if _probe is None:
        path = PKG_DIR.parent / "scripts" / "libservingprobe.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib()                                   # (libvkindex first: the probe links against it)
        P = C.CDLL(str(path))
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        P.vk_probe_submit.argtypes = [vp, vp, u64, u32, u64, u64, i32, i32, u64, vp, vp, C.POINTER(ProbeResult)]
        P.vk_probe_blocking.argtypes = [vp, vp, u64, u32, u64, u64, i32, i32, vp, vp, C.POINTER(ProbeResult)]
        P.vk_probe_add_single.argtypes = [vp, vp, vp, u64, u32, i32, u64, C.POINTER(C.c_double)]
        P.vk_probe_add_single.restype = u64
        _probe = P
    return _probe


def probe_submit(ix, Q, k, total, producers=4, window=1024, ef=0, ref=None):
    """`producers` native threads keep `window` vk_index_search_submit requests outstanding until `total` have completed;
    ref = (dist [nq][k] f32, labels [nq][k] u64) to compare every answer with"""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    rd = rl = None
    if ref is not None:
        rd, rl = np.ascontiguousarray(ref[0], dtype=np.float32), np.ascontiguousarray(ref[1], dtype=np.uint64)
    r = ProbeResult()
    _check(probe_lib().vk_probe_submit(ix._h, _ptr(Q), Q.shape[0], Q.shape[1], k, ef, producers, window, total, _ptr(rd), _ptr(rl), C.byref(r)))
    return r


def probe_add_single(ix, rows, labels=None, threads=16, grow_by=0):
    """n rows through vk_index_add, one call each, from `threads` native writer threads (IndexSchema's AddRecord traffic);
    returns (failed adds, seconds)"""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
    sec = C.c_double()
    failed = probe_lib().vk_probe_add_single(ix._h, _ptr(lab), _ptr(rows), rows.shape[0], rows.shape[1], threads, grow_by, C.byref(sec))
    return int(failed), sec.value


def probe_blocking(ix, Q, k, threads, calls, ef=0, ref=None):
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    rd = rl = None
    if ref is not None:
        rd, rl = np.ascontiguousarray(ref[0], dtype=np.float32), np.ascontiguousarray(ref[1], dtype=np.uint64)
    r = ProbeResult()
    _check(probe_lib().vk_probe_blocking(ix._h, _ptr(Q), Q.shape[0], Q.shape[1], k, ef, threads, calls, _ptr(rd), _ptr(rl), C.byref(r)))
    return r


_aprobe = None


def adaptor_probe(ix, Q, k, total, readers, window, ef=0, hnsw=False, m=16, blocking=False, max_batch=0, wait_us=0, ref=None, fronts=1):
    """scripts/adaptor_probe.cc: single-query traffic through VectorGpuFlat / VectorGpuHNSW (include/vk_vector_adaptor.h over the
    mock of VectorBase) adopted onto `ix`: `window` FT.SEARCHes outstanding, a reader pool of `readers` threads, SearchAsync
    (or, blocking=True, Search), `fronts` threads feeding the pool.  max_batch = 0: the coalescing the adaptor sets itself."""
    global _aprobe
    if _aprobe is None:
        path = PKG_DIR.parent / "scripts" / "libadaptorprobe.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib()
        P = C.CDLL(str(path))
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        P.vk_adaptor_probe.argtypes = [vp, i32, u32, u32, vp, u64, u64, u64, i32, i32, i32, u64, i32, u32, u32, vp, vp, C.POINTER(ProbeResult)]
        _aprobe = P
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    rd = rl = None
    if ref is not None:
        rd, rl = np.ascontiguousarray(ref[0], dtype=np.float32), np.ascontiguousarray(ref[1], dtype=np.uint64)
    r = ProbeResult()
    _check(_aprobe.vk_adaptor_probe(ix._h, int(hnsw), Q.shape[1], int(m), _ptr(Q), Q.shape[0], k, ef, readers, int(fronts), window, total, int(blocking),
                                    int(max_batch), int(wait_us), _ptr(rd), _ptr(rl), C.byref(r)))
    return r
