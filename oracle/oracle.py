"""ctypes front-end of the CPU ORACLE (oracle/liboracle.so) and of the compiled
reference SimSIMD (oracle/_ref/libsimsimd_ref.so, when present).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ISA = {"serial": 0, "haswell": 1, "skylake": 2}
SPACE = {"L2": 0, "IP": 1, "COSINE": 1}

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (always) and _ref (only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", str(HERE), "liboracle.so"])
    if ref and Path("/root/reference/third_party/simsimd/c/lib.c").exists():
        subprocess.check_call(["make", "-s", "-C", str(HERE), "ref"], stderr=subprocess.DEVNULL)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _load():
    so = HERE / "liboracle.so"
    if not so.exists():
        build(ref=False)
    lib = C.CDLL(str(so))
    lib.vko_dot_f32.restype = C.c_double
    lib.vko_dot_f32.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
    lib.vko_l2sq_f32.restype = C.c_double
    lib.vko_l2sq_f32.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
    lib.vko_distance.restype = C.c_float
    lib.vko_distance.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_size_t]
    lib.vko_normalize.restype = C.c_float
    lib.vko_normalize.argtypes = [_f32p, _f32p, C.c_size_t]
    lib.vko_cpu_path.restype = C.c_char_p
    lib.vko_last_error.restype = C.c_char_p
    lib.vko_flat_new.restype = C.c_void_p
    lib.vko_flat_new.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t]
    lib.vko_flat_free.argtypes = [C.c_void_p]
    for n in ("vko_flat_add", "vko_flat_add_borrowed"):
        getattr(lib, n).restype = C.c_int
        getattr(lib, n).argtypes = [C.c_void_p, _f32p, C.c_uint64]
    lib.vko_flat_add_many.restype = C.c_int
    lib.vko_flat_add_many.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.vko_flat_remove.argtypes = [C.c_void_p, C.c_uint64]
    lib.vko_flat_resize.argtypes = [C.c_void_p, C.c_size_t]
    lib.vko_flat_count.restype = C.c_size_t
    lib.vko_flat_count.argtypes = [C.c_void_p]
    lib.vko_flat_capacity.restype = C.c_size_t
    lib.vko_flat_capacity.argtypes = [C.c_void_p]
    lib.vko_flat_search.restype = C.c_size_t
    lib.vko_flat_search.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u64p, C.c_uint64, C.c_long, _f32p, _u64p]
    lib.vko_flat_distance.restype = C.c_int
    lib.vko_flat_distance.argtypes = [C.c_void_p, C.c_uint64, _f32p, _f32p]
    lib.vko_prefilter_topk.restype = C.c_size_t
    lib.vko_prefilter_topk.argtypes = [C.c_int, C.c_int, C.c_size_t, _f32p, C.POINTER(_f32p), _u64p,
                                       C.c_size_t, C.c_size_t, _f32p, _u64p]
    lib.vko_hnsw_new.restype = C.c_void_p
    lib.vko_hnsw_new.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    lib.vko_hnsw_free.argtypes = [C.c_void_p]
    lib.vko_hnsw_set_ef.argtypes = [C.c_void_p, C.c_size_t]
    lib.vko_hnsw_add.restype = C.c_int
    lib.vko_hnsw_add.argtypes = [C.c_void_p, _f32p, C.c_uint64]
    lib.vko_hnsw_add_into.restype = C.c_int
    lib.vko_hnsw_add_into.argtypes = [C.c_void_p, _f32p, C.c_uint64, C.c_uint32]
    lib.vko_hnsw_vacant.restype = C.c_size_t
    lib.vko_hnsw_vacant.argtypes = [C.c_void_p, _u32p, C.c_size_t]
    lib.vko_hnsw_mark_delete.restype = C.c_int
    lib.vko_hnsw_mark_delete.argtypes = [C.c_void_p, C.c_uint64]
    lib.vko_hnsw_resize.argtypes = [C.c_void_p, C.c_size_t]
    for n in ("vko_hnsw_count", "vko_hnsw_deleted_count", "vko_hnsw_capacity"):
        getattr(lib, n).restype = C.c_size_t
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.vko_hnsw_max_level.restype = C.c_int
    lib.vko_hnsw_max_level.argtypes = [C.c_void_p]
    lib.vko_hnsw_entry_point.restype = C.c_uint32
    lib.vko_hnsw_entry_point.argtypes = [C.c_void_p]
    lib.vko_hnsw_search.restype = C.c_size_t
    lib.vko_hnsw_search.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, _u64p, C.c_uint64, C.c_long,
                                    _f32p, _u64p, _u64p, _u64p]
    lib.vko_hnsw_distance.restype = C.c_int
    lib.vko_hnsw_distance.argtypes = [C.c_void_p, C.c_uint64, _f32p, _f32p]
    lib.vko_hnsw_level_of.restype = C.c_int
    lib.vko_hnsw_level_of.argtypes = [C.c_void_p, C.c_uint32]
    lib.vko_hnsw_label_of.restype = C.c_uint64
    lib.vko_hnsw_label_of.argtypes = [C.c_void_p, C.c_uint32]
    lib.vko_hnsw_is_deleted.restype = C.c_int
    lib.vko_hnsw_is_deleted.argtypes = [C.c_void_p, C.c_uint32]
    lib.vko_hnsw_links.restype = C.c_size_t
    lib.vko_hnsw_links.argtypes = [C.c_void_p, C.c_uint32, C.c_int, _u32p, C.c_size_t]
    lib.vko_hnsw_row.restype = _f32p
    lib.vko_hnsw_row.argtypes = [C.c_void_p, C.c_uint32]
    lib.vko_hnsw_load_graph.restype = C.c_int
    lib.vko_hnsw_load_graph.argtypes = [C.c_void_p, C.c_size_t, _f32p, _u64p, _u32p, _u64p, _u32p, C.c_int, C.c_uint32]
    lib.vko_hnsw_view.restype = C.c_void_p
    lib.vko_hnsw_view.argtypes = [C.c_void_p]
    lib.vko_sink_new.restype = C.c_void_p
    lib.vko_sink_new.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t]
    lib.vko_sink_write.restype = C.c_int
    lib.vko_sink_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.vko_sink_finish.restype = C.c_void_p
    lib.vko_sink_finish.argtypes = [C.c_void_p]
    lib.vko_msink_new.restype = C.c_void_p
    lib.vko_msink_new.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t]
    lib.vko_msink_write.restype = C.c_int
    lib.vko_msink_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.vko_msink_count.restype = C.c_uint64
    lib.vko_msink_count.argtypes = [C.c_void_p]
    lib.vko_msink_take.restype = C.c_void_p
    lib.vko_msink_take.argtypes = [C.c_void_p, C.c_uint64]
    lib.vko_msink_free.restype = None
    lib.vko_msink_free.argtypes = [C.c_void_p]
    lib.vko_merge_topk.restype = C.c_size_t
    lib.vko_merge_topk.argtypes = [_f32p, _u64p, _u32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, _u64p]
    return lib


LIB = _load()


def cpu_path() -> str:
    return LIB.vko_cpu_path().decode()


def last_error() -> str:
    return LIB.vko_last_error().decode()


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def dot(a, b, isa="skylake") -> float:
    a, b = f32(a), f32(b)
    return LIB.vko_dot_f32(ISA[isa], _fp(a), _fp(b), a.size)


def l2sq(a, b, isa="skylake") -> float:
    a, b = f32(a), f32(b)
    return LIB.vko_l2sq_f32(ISA[isa], _fp(a), _fp(b), a.size)


def distance(space, a, b, isa="skylake") -> np.float32:
    a, b = f32(a), f32(b)
    return np.float32(LIB.vko_distance(SPACE[space], ISA[isa], _fp(a), _fp(b), a.size))


def normalize(v):
    v = f32(v)
    out = np.empty_like(v)
    mag = LIB.vko_normalize(_fp(out), _fp(v), v.size)
    return out, np.float32(mag)


def allow_bitmap(labels, nbits: int) -> np.ndarray:
    """Bitmap indexed by label (bit set = allowed)."""
    bits = np.zeros((nbits + 63) // 64, dtype=np.uint64)
    labels = np.asarray(labels, dtype=np.uint64)
    np.bitwise_or.at(bits, (labels >> np.uint64(6)).astype(np.int64),
                     np.uint64(1) << (labels & np.uint64(63)))
    return bits


def _allow_args(allow, nbits):
    if allow is None:
        return None, 0
    allow = np.ascontiguousarray(allow, dtype=np.uint64)
    return allow.ctypes.data_as(_u64p), int(nbits if nbits is not None else allow.size * 64)


class Flat:
    """hnswlib::BruteforceSearch<float> restated (bruteforce.h)."""

    def __init__(self, dim, space="L2", isa="skylake", max_elements=1024):
        self.dim, self.space = dim, space
        self._h = LIB.vko_flat_new(dim, SPACE[space], ISA[isa], max_elements)
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            LIB.vko_flat_free(self._h)
            self._h = None

    def add(self, row, label) -> int:
        row = f32(row)
        assert row.size == self.dim
        return LIB.vko_flat_add(self._h, _fp(row), int(label))

    def add_many(self, rows, labels=None, borrowed=False):
        rows = f32(rows)
        if borrowed:
            self._keep.append(rows)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
        rc = LIB.vko_flat_add_many(self._h, rows.ctypes.data, rows.strides[0], None if lab is None else lab.ctypes.data,
                                   rows.shape[0], int(borrowed))
        if rc:
            raise RuntimeError(last_error())

    def use_reference_distance(self) -> bool:
        """distances by the COMPILED REFERENCE's fstdistfunc_ (oracle/_ref/libsimsimd_ref.so: third_party/hnswlib/simsimd.h over
        SimSIMD as the reference builds it) instead of the restated kernels; False when _ref is not there"""
        if not Ref.available():
            return False
        ref = Ref()
        fn = ref.lib.ref_L2SqrSimsimd if self.space == "L2" else ref.lib.ref_InnerProductDistanceSimsimd
        LIB.vko_flat_set_distfn.argtypes = [C.c_void_p, C.c_void_p]
        LIB.vko_flat_set_distfn.restype = None
        LIB.vko_flat_set_distfn(self._h, C.cast(fn, C.c_void_p))
        self._keep.append(ref)
        return True

    def remove(self, label):
        LIB.vko_flat_remove(self._h, int(label))

    def resize(self, n):
        LIB.vko_flat_resize(self._h, n)

    @property
    def count(self):
        return LIB.vko_flat_count(self._h)

    @property
    def capacity(self):
        return LIB.vko_flat_capacity(self._h)

    def search(self, q, k, allow=None, allow_nbits=None, cancel_after=-1):
        q = f32(q)
        od = np.empty(max(k, 1), dtype=np.float32)
        ol = np.empty(max(k, 1), dtype=np.uint64)
        ap, nb = _allow_args(allow, allow_nbits)
        n = LIB.vko_flat_search(self._h, _fp(q), k, ap, nb, cancel_after, _fp(od), ol.ctypes.data_as(_u64p))
        return od[:n].copy(), ol[:n].copy()

    def distance(self, label, q):
        q = f32(q)
        out = C.c_float()
        rc = LIB.vko_flat_distance(self._h, int(label), _fp(q), C.byref(out))
        return None if rc else np.float32(out.value)


class HNSW:
    """hnswlib::HierarchicalNSW<float> restated (hnswalg.h), single-threaded."""

    def __init__(self, dim, space="L2", isa="skylake", max_elements=1024, M=16, ef_construction=200,
                 seed=100, allow_replace_deleted=False, ef=10):
        self.dim, self.space, self.M = dim, space, M
        self._h = LIB.vko_hnsw_new(dim, SPACE[space], ISA[isa], max_elements, M, ef_construction, seed,
                                   int(allow_replace_deleted))
        LIB.vko_hnsw_set_ef(self._h, ef)

    def __del__(self):
        if getattr(self, "_h", None):
            LIB.vko_hnsw_free(self._h)
            self._h = None

    def add(self, row, label) -> int:
        row = f32(row)
        assert row.size == self.dim
        return LIB.vko_hnsw_add(self._h, _fp(row), int(label))

    def add_into(self, row, label, slot) -> int:
        """addPoint(replace_deleted) with the tombstoned slot to reuse named by the caller (hnswalg.h:1306-1309 takes
        `*deleted_elements.begin()` of an unordered_set: the choice is the container's, not the algorithm's).
        2 if `slot` is not vacant; a known label or no vacancy: plain add."""
        row = f32(row)
        assert row.size == self.dim
        return LIB.vko_hnsw_add_into(self._h, _fp(row), int(label), int(slot))

    def vacant(self):
        """The tombstoned slots a new label may take over (deleted_elements)."""
        n = LIB.vko_hnsw_vacant(self._h, None, 0)
        out = np.empty(max(n, 1), np.uint32)
        LIB.vko_hnsw_vacant(self._h, out.ctypes.data_as(_u32p), n)
        return out[:n].tolist()

    def add_many(self, rows, labels=None):
        rows = f32(rows)
        base, stride = rows.ctypes.data, rows.strides[0]
        for i in range(rows.shape[0]):
            lab = int(labels[i]) if labels is not None else i
            rc = LIB.vko_hnsw_add(self._h, C.cast(base + i * stride, _f32p), lab)
            if rc:
                raise RuntimeError(last_error())

    def mark_delete(self, label) -> int:
        return LIB.vko_hnsw_mark_delete(self._h, int(label))

    def resize(self, n):
        LIB.vko_hnsw_resize(self._h, n)

    def set_ef(self, ef):
        LIB.vko_hnsw_set_ef(self._h, ef)

    count = property(lambda s: LIB.vko_hnsw_count(s._h))
    deleted_count = property(lambda s: LIB.vko_hnsw_deleted_count(s._h))
    capacity = property(lambda s: LIB.vko_hnsw_capacity(s._h))
    max_level = property(lambda s: LIB.vko_hnsw_max_level(s._h))
    entry_point = property(lambda s: LIB.vko_hnsw_entry_point(s._h))

    def search(self, q, k, ef=0, allow=None, allow_nbits=None, cancel_after=-1, stats=False):
        q = f32(q)
        od = np.empty(max(k, 1), dtype=np.float32)
        ol = np.empty(max(k, 1), dtype=np.uint64)
        ne, nh = C.c_uint64(), C.c_uint64()
        ap, nb = _allow_args(allow, allow_nbits)
        n = LIB.vko_hnsw_search(self._h, _fp(q), k, ef, ap, nb, cancel_after, _fp(od),
                                ol.ctypes.data_as(_u64p), C.byref(ne), C.byref(nh))
        if stats:
            return od[:n].copy(), ol[:n].copy(), ne.value, nh.value
        return od[:n].copy(), ol[:n].copy()

    def distance(self, label, q):
        q = f32(q)
        out = C.c_float()
        rc = LIB.vko_hnsw_distance(self._h, int(label), _fp(q), C.byref(out))
        return None if rc else np.float32(out.value)

    @classmethod
    def _adopt(cls, handle, dim, space, M, base=None):
        self = cls.__new__(cls)
        self.dim, self.space, self.M, self._h, self._base = dim, space, M, handle, base
        return self

    def view(self):
        """One more searching thread over the same graph: own visited list and counters (hnswlib gives every
        concurrent search its own VisitedList); keeps the base alive."""
        return HNSW._adopt(LIB.vko_hnsw_view(self._h), self.dim, self.space, self.M, base=self)

    @classmethod
    def from_product_index(cls, save_fn, dim, space, M, ef_construction=200, isa="skylake", ef=10):
        """The graph a PRODUCT index holds, through its own SaveIndex chunk stream, chunk by chunk in C:
        `save_fn(callback, user)` must call vk_index_save(ix, callback, user).  10M elements pass without one Python
        object per chunk."""
        sink = LIB.vko_sink_new(dim, SPACE[space], ISA[isa], M, ef_construction)
        cb = C.cast(LIB.vko_sink_write, C.c_void_p)
        rc = save_fn(cb, C.c_void_p(sink))
        h = LIB.vko_sink_finish(sink)
        if rc or not h:
            if h:
                LIB.vko_hnsw_free(h)
            raise RuntimeError(f"save into the oracle sink failed (rc={rc}): {last_error()}")
        self = cls._adopt(h, dim, space, M)
        LIB.vko_hnsw_set_ef(self._h, ef)
        return self

    @classmethod
    def shards_from_product_index(cls, save_fn, dim, space, M, ef_construction=200, isa="skylake", ef=10):
        """The graphs of a SHARDED product index (vk_index_params.n_shards >= 1), one oracle graph per shard, from its
        own save stream (a marker chunk, then every shard's SaveIndex stream): the CPU side of a sharded-search parity
        check searches each shard's very graph and merges with merge_topk."""
        ms = LIB.vko_msink_new(dim, SPACE[space], ISA[isa], M, ef_construction)
        try:
            rc = save_fn(C.cast(LIB.vko_msink_write, C.c_void_p), C.c_void_p(ms))
            n = LIB.vko_msink_count(ms)
            if rc or not n:
                raise RuntimeError(f"save into the oracle shard sink failed (rc={rc}): {last_error()}")
            out = []
            for i in range(n):
                g = cls._adopt(LIB.vko_msink_take(ms, i), dim, space, M)
                LIB.vko_hnsw_set_ef(g._h, ef)
                out.append(g)
            return out
        finally:
            LIB.vko_msink_free(ms)

    @classmethod
    def from_saved_chunks(cls, chunks, dim, space, M, ef_construction=200, isa="skylake", ef=10):
        """Oracle index over a graph serialised in hnswlib's SaveIndex chunk layout (hnswalg.h:808-865):
        chunks[0] header, then n element chunks [level-0 words | row | label], then per element a u64
        size chunk and (if non-zero) the upper-level link block."""
        sl0 = (2 * M + 1) * 4
        esz = sl0 + dim * 4 + 8
        n = 0
        while 1 + n < len(chunks) and len(chunks[1 + n]) == esz and not (len(chunks[1 + n]) == 8):
            n += 1
        # the element section ends where the size chunks (8 bytes) begin; esz == 8 is impossible
        body = np.frombuffer(b"".join(chunks[1:1 + n]), dtype=np.uint8).reshape(n, esz) if n else np.zeros((0, esz), np.uint8)
        l0 = np.ascontiguousarray(body[:, :sl0]).view(np.uint32).reshape(n, 2 * M + 1)
        rows = np.ascontiguousarray(body[:, sl0:sl0 + dim * 4]).view(np.float32).reshape(n, dim)
        labels = np.ascontiguousarray(body[:, sl0 + dim * 4:]).view(np.uint64).reshape(n)
        off = np.zeros(n + 1, np.uint64)
        ups = []
        pos = 1 + n
        for i in range(n):
            sz = int(np.frombuffer(chunks[pos], dtype=np.uint64)[0])
            pos += 1
            if sz:
                ups.append(np.frombuffer(chunks[pos], dtype=np.uint32))
                pos += 1
            off[i + 1] = off[i] + sz // 4
        up = np.ascontiguousarray(np.concatenate(ups)) if ups else np.zeros(1, np.uint32)
        levels = ((off[1:] - off[:-1]) // np.uint64(M + 1)).astype(np.int64)
        max_level = int(levels.max()) if n else -1
        # entry point: the header's field 8 (varint); parse the few fields we need
        hdr = chunks[0]
        fields, p = {}, 0
        while p < len(hdr):
            key = hdr[p]; p += 1
            f, w = key >> 3, key & 7
            if w == 0:
                v, sh = 0, 0
                while True:
                    b = hdr[p]; p += 1
                    v |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                fields[f] = v
            elif w == 1:
                p += 8
        self = cls(dim, space, isa=isa, max_elements=max(n, 1), M=M, ef_construction=ef_construction, ef=ef)
        rc = LIB.vko_hnsw_load_graph(self._h, n, _fp(rows), labels.ctypes.data_as(_u64p), l0.ctypes.data_as(_u32p),
                                     off.ctypes.data_as(_u64p), up.ctypes.data_as(_u32p), max_level,
                                     int(fields.get(8, 0)) & 0xFFFFFFFF)
        if rc:
            raise RuntimeError(last_error())
        return self

    def export_graph(self):
        """Dense arrays describing the graph (for feeding the device path the same graph)."""
        n = self.count
        levels = np.array([LIB.vko_hnsw_level_of(self._h, i) for i in range(n)], dtype=np.int32)
        labels = np.array([LIB.vko_hnsw_label_of(self._h, i) for i in range(n)], dtype=np.uint64)
        deleted = np.array([LIB.vko_hnsw_is_deleted(self._h, i) for i in range(n)], dtype=np.uint8)
        maxm0 = 2 * self.M
        l0 = np.zeros((n, maxm0 + 1), dtype=np.uint32)
        buf = (C.c_uint32 * (maxm0 + 1))()
        upper = {}
        for i in range(n):
            c = LIB.vko_hnsw_links(self._h, i, 0, buf, maxm0)
            l0[i, 0] = c
            l0[i, 1:1 + c] = np.frombuffer(buf, dtype=np.uint32, count=c)
            for lv in range(1, levels[i] + 1):
                c = LIB.vko_hnsw_links(self._h, i, lv, buf, self.M)
                upper[(i, lv)] = np.frombuffer(buf, dtype=np.uint32, count=c).copy()
        rows = np.stack([np.ctypeslib.as_array(LIB.vko_hnsw_row(self._h, i), shape=(self.dim,)).copy()
                         for i in range(n)]) if n else np.zeros((0, self.dim), np.float32)
        return dict(levels=levels, labels=labels, deleted=deleted, l0=l0, upper=upper, rows=rows,
                    entry_point=self.entry_point, max_level=self.max_level)


def prefilter_topk(space, q, rows, labels, k, isa="skylake"):
    q, rows = f32(q), f32(rows)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    n = rows.shape[0]
    ptrs = (_f32p * max(n, 1))()
    for i in range(n):
        ptrs[i] = C.cast(rows.ctypes.data + i * rows.strides[0], _f32p)
    od = np.empty(max(k, 1), dtype=np.float32)
    ol = np.empty(max(k, 1), dtype=np.uint64)
    m = LIB.vko_prefilter_topk(SPACE[space], ISA[isa], rows.shape[1] if n else q.size, _fp(q), ptrs,
                               labels.ctypes.data_as(_u64p), n, k, _fp(od), ol.ctypes.data_as(_u64p))
    return od[:m].copy(), ol[:m].copy()


def merge_topk(dist, label, counts, k):
    dist = f32(dist)
    label = np.ascontiguousarray(label, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    parts, per = dist.shape
    od = np.empty(max(k, 1), dtype=np.float32)
    ol = np.empty(max(k, 1), dtype=np.uint64)
    n = LIB.vko_merge_topk(_fp(dist), label.ctypes.data_as(_u64p), counts.ctypes.data_as(_u32p), parts, per, k,
                           _fp(od), ol.ctypes.data_as(_u64p))
    return od[:n].copy(), ol[:n].copy()


# ---- the compiled reference SimSIMD (present only after `make ref`) -------------
class Ref:
    """oracle/_ref/libsimsimd_ref.so: the reference's own simsimd c/lib.c + export shims."""

    def __init__(self):
        so = HERE / "_ref" / "libsimsimd_ref.so"
        if not so.exists():
            raise FileNotFoundError(so)
        self.lib = C.CDLL(str(so))
        for n in ("dot", "l2sq"):
            for isa in ("serial", "haswell", "skylake"):
                fn = getattr(self.lib, f"ref_{n}_f32_{isa}")
                fn.argtypes = [_f32p, _f32p, C.c_size_t, C.POINTER(C.c_double)]
                fn.restype = None
            fn = getattr(self.lib, f"simsimd_{n}_f32")
            fn.argtypes = [_f32p, _f32p, C.c_size_t, C.POINTER(C.c_double)]
            fn.restype = None
        self.lib.ref_InnerProductDistanceSimsimd.restype = C.c_float
        self.lib.ref_InnerProductDistanceSimsimd.argtypes = [_f32p, _f32p, C.c_size_t]
        self.lib.ref_L2SqrSimsimd.restype = C.c_float
        self.lib.ref_L2SqrSimsimd.argtypes = [_f32p, _f32p, C.c_size_t]
        self.lib.ref_capabilities.restype = C.c_uint

    @staticmethod
    def available() -> bool:
        return (HERE / "_ref" / "libsimsimd_ref.so").exists()

    def capabilities(self) -> int:
        return self.lib.ref_capabilities()

    def kernel(self, name, isa, a, b) -> float:
        a, b = f32(a), f32(b)
        out = C.c_double()
        fn = getattr(self.lib, f"simsimd_{name}_f32" if isa == "dispatch" else f"ref_{name}_f32_{isa}")
        fn(_fp(a), _fp(b), a.size, C.byref(out))
        return out.value

    def distance(self, space, a, b) -> np.float32:
        a, b = f32(a), f32(b)
        fn = self.lib.ref_InnerProductDistanceSimsimd if SPACE[space] == 1 else self.lib.ref_L2SqrSimsimd
        return np.float32(fn(_fp(a), _fp(b), a.size))
