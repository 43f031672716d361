"""The visited set of a batched HNSW search as an exact hash set of node ids.

hnswlib marks visited nodes in a tag array sized by the graph (visited_list_pool.h; hnswalg.h:453-464).  The device
keeps one visited set per resident wave; on a large graph a bitmap of all nodes per wave (1.25 MB at 10M) limits how many
waves can run and is mostly cleared, never touched.  A batch that fills the device therefore uses an open-addressing
table of the ids a search actually touches; a query that would fill its table beyond 3/4 is abandoned and re-run by a
second launch with the bitmap.  Whatever the path: ids, distance bits and the layer-0 work counters are the oracle's on
the SAME graph."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


class _Env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _check(g, o, Q, k, ef):
    D, L, N = g.search_batch(Q, k, ef=ef)
    ne = nh = 0
    for i in range(len(Q)):
        od, ol, e, h = o.search(Q[i], k, ef=ef, stats=True)
        assert L[i, :N[i]].tolist() == ol.tolist(), i
        assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), i
        ne += e
        nh += h
    st = g.stats()
    assert (st.last_n_eval, st.last_n_hops) == (ne, nh)
    assert st.last_frontier_dropped == 0
    return st


@pytest.mark.parametrize("metric", ["L2", "COSINE"])
@pytest.mark.parametrize("ef,k", [(32, 10), (128, 10), (300, 40), (700, 20)])   # register lists (1, 2, 8 slots) and the LDS list
@pytest.mark.parametrize("log2", [None, 7, 9])   # table sized by ef / 128 words (nearly every query re-run) / 512 words (some)
def test_hash_visited_set_matches_the_oracle(vsa, oracle, metric, ef, k, log2):
    rng = np.random.default_rng(97 + ef)
    n, dim, M = 5000, 48, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    env = {"VK_HNSW_VISITED_HASH": 2, "VK_HNSW_VISITED_MODE": 0}     # (the table in memory: r04's default keeps small searches' sets in LDS)
    if log2:
        env["VK_HNSW_HASH_LOG2"] = log2
    with _Env(**env):
        g = vsa.Index("HNSW", dim, metric, initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, metric, M, ef_construction=40)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    st = _check(g, o, Q, k, ef)
    if log2 == 7:
        assert st.last_frontier_redo > 0       # 96 entries hold no search with ef >= 32 on this graph
    if log2 is None:
        assert st.last_frontier_redo == 0      # sized by ef: nobody outgrows it here


def test_hash_visited_set_bf16_rows(vsa, oracle):
    """bf16 rows take the same code (the reference only has FLOAT32: parity against the oracle over the rounded rows,
    graph built point by point on both sides)."""
    rng = np.random.default_rng(11)
    n, dim = 2500, 64
    x = rng.standard_normal((n, dim)).astype(np.float32)
    with _Env(VK_HNSW_VISITED_HASH=2, VK_HNSW_HASH_LOG2=9, VK_HNSW_VISITED_MODE=0):
        g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, dtype="bf16")
    for i in range(n):
        assert g.add(i, x[i]) == 0
    u = x.view(np.uint32)
    xr = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    o = oracle.HNSW(dim, "L2", max_elements=n, M=16, ef_construction=100)
    o.add_many(xr)
    Q = rng.standard_normal((40, dim)).astype(np.float32)
    st = _check(g, o, Q, 10, 64)
    assert st.last_frontier_redo > 0


def test_default_choice_small_graph_keeps_the_bitmap_and_filters_never_hash(vsa, oracle):
    rng = np.random.default_rng(3)
    n, dim, M = 4000, 32, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    Q = rng.standard_normal((600, dim)).astype(np.float32)     # a batch past the latency variant's limit
    st = _check(g, o, Q[:600:10], 10, 64)
    assert st.last_frontier_redo == 0
    # forced on, but with a filter: the HBM-frontier kernels keep their bitmaps, answers as before
    with _Env(VK_HNSW_VISITED_HASH=2, VK_HNSW_HASH_LOG2=7):
        g2 = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g2.add_batch(x)
    g2.flush()
    o2 = oracle.HNSW.from_product_index(g2.save_raw, dim, "L2", M, ef_construction=40)
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.3), n)
    D, L, N = g2.search_batch(Q[:30], 10, ef=64, allow=bits, allow_nbits=n)
    for i in range(30):
        od, ol = o2.search(Q[i], 10, ef=64, allow=bits, allow_nbits=n)
        assert L[i, :N[i]].tolist() == ol.tolist()
        assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("mode", [0, 1, 2, 4])
@pytest.mark.parametrize("ef,k,log2", [(32, 10, None), (128, 10, None), (300, 40, None), (512, 10, None), (544, 10, None), (700, 20, None),
                                       (64, 10, 7), (128, 10, 9)])   # (512 / 544: the LDS set with the trimmed frontier, 8 / 16 list slots per lane)
def test_other_ways_of_keeping_the_hash_set_match_the_oracle(vsa, oracle, mode, ef, k, log2):
    """r04, option hnsw-visited-mode: 0 = compare-and-swap at agent scope (r02), 1 = at WAVEFRONT scope (the set is private
    to its wave), 2 = the table in buckets of eight ids whose fill counts live in LDS -- a look-up is a 32-byte read (none
    at all when the bucket is empty), an insert a store nobody waits for, no atomic ever touches memory.  Measured on the
    10M graph (profiles/r04_hnsw_visited_modes.log) these three run within 3 % of each other: what the visited set costs
    is its accesses' place in the memory system's mix, not how they are made.  4 = the whole set in LDS whenever it fits
    (two-choice buckets of six 16-bit entries; the default, 3, takes it only where a search is sure to stay below its
    4800 ids): +24 % at ef = 128 on the 10M graph.  Same ids, distance bits and work counters as the oracle on the same
    graph, including the queries a small table -- or an outgrown LDS set -- sends to the second launch."""
    rng = np.random.default_rng(1000 + ef + mode)
    n, dim, M = 5000, 48, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    env = {"VK_HNSW_VISITED_HASH": 2, "VK_HNSW_VISITED_MODE": mode}
    if log2:
        env["VK_HNSW_HASH_LOG2"] = log2
    # (one builder thread: the same graph every run.  With several the graph differs from build to build, and about one
    #  (graph, query) in tens of thousands has two frontier candidates at EXACTLY the same f32 distance whose expansion order
    #  decides whether the second is still expanded -- hnswlib's heaps compare distances only, so its order is libstdc++'s sift
    #  order; the kernels take the smaller pool index: same answer, one hop more or less.  DESIGN.md section 2 states the contract for tie-free data; scripts/
    #  hnsw_hop_mismatch_hunt.py finds such a case, profiles/r05_hnsw_tie_hop.log.)
    with _Env(**env):
        g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=1)
    assert g.get_option("hnsw-visited-mode") == mode
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    st = _check(g, o, Q, k, ef)
    if log2 == 7 and (mode != 4 or ef > 256):      # (mode 4 keeps the set in LDS while the result list is in registers: no table in memory to be small)
        assert st.last_frontier_redo > 0
    if log2 is None:
        assert st.last_frontier_redo == 0
    # the mode is a run-time option: to the default (3: the LDS set where a search is sure to fit it) on the same index
    g.set_option("hnsw-visited-mode", 3)
    _check(g, o, Q[:20], k, ef)


def test_a_search_that_outgrows_the_lds_set_spills_into_memory(vsa, oracle):
    """hnsw-visited-mode 4 takes the LDS set (6144 slots in 12 KB) wherever it fits the LDS; on a graph with 64 links per node
    a search at ef = 256 evaluates about 5000 nodes, at ef = 400 (the 32 KB set: 16384 slots) about 7500.  An id whose four
    candidate buckets are full goes to the wave's table in memory instead and is found there next time -- nothing is
    re-run, and ids, distance bits and work counters are the oracle's all the same.  (The default, mode 3, takes an LDS set
    only where ef x maxM0 keeps a search well inside it.)"""
    rng = np.random.default_rng(41)
    n, dim, M = 12000, 48, 32
    x = rng.standard_normal((n, dim)).astype(np.float32)
    with _Env(VK_HNSW_VISITED_HASH=2, VK_HNSW_VISITED_MODE=4):
        g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=60, build_threads=8)
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=60)
    Q = rng.standard_normal((60, dim)).astype(np.float32)
    for ef in (256, 400, 48):                # the 12 KB set overfull, the 32 KB set, a search either holds with room to spare
        st = _check(g, o, Q, 10, ef)
        assert st.last_frontier_redo == 0
    g.set_option("hnsw-visited-mode", 3)     # the default rule: 256 x 64 is beyond both LDS sets' budgets -> the table in memory
    st = _check(g, o, Q, 10, 256)
    assert st.last_frontier_redo == 0


@pytest.mark.parametrize("mode,log2", [(0, None), (4, None), (0, 7)])
def test_a_few_tombstones_keep_the_frontier_on_chip(vsa, oracle, mode, log2):
    """r06: one deleted key used to send EVERY search of an index to the HBM-frontier kernel (0.58 of the HBM peak against 0.74
    at 1.25M x 768, no filter involved).  Up to 1 / 16 of the nodes tombstoned and no filter: the batch takes the launch with the
    frontier in LDS and the visited set on chip; a query that outgrows either is re-run by the launch nothing can overflow
    (log2 = 7: nearly all of them).  Tombstone semantics (hnswalg.h:373-388, :515-524) are the shared kernel body's: ids,
    distance bits and work counters are the oracle's on the same graph; beyond 1 / 16, with the option off, or with a filter
    the HBM frontier is back."""
    rng = np.random.default_rng(4100 + mode)
    n, dim, M = 6000, 48, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    env = {"VK_HNSW_VISITED_HASH": 2, "VK_HNSW_VISITED_MODE": mode}
    if log2:
        env["VK_HNSW_HASH_LOG2"] = log2
    with _Env(**env):
        g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=1)
    g.add_batch(x)
    dead = rng.choice(n, n // 100, replace=False)
    for lab in dead:
        assert g.remove(int(lab)) == 0
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    assert o.deleted_count == len(dead)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    for ef, k in ((64, 10), (300, 40)):
        st = _check(g, o, Q, k, ef)
        assert st.last_visited_mode != 0, "a few tombstones: the optimistic launch"
        assert (st.last_frontier_redo > 0) == (log2 == 7)
        L = g.search_batch(Q, k, ef=ef)[1]
        assert not set(L.ravel().tolist()) & set(int(v) for v in dead)
    # the option off: the HBM frontier, same answers
    g.set_option("hnsw-optimistic-tombstones", 0)
    assert _check(g, o, Q, 10, 64).last_visited_mode == 0
    g.set_option("hnsw-optimistic-tombstones", 1)
    # a filter on top: HBM frontier (the frontier grows like 1 / selectivity)
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < 0.3), n)
    D, L, N = g.search_batch(Q[:20], 10, ef=64, allow=bits, allow_nbits=n)
    assert g.stats().last_visited_mode == 0
    for i in range(20):
        od, ol = o.search(Q[i], 10, ef=64, allow=bits, allow_nbits=n)
        assert L[i, :N[i]].tolist() == ol.tolist() and D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist()
    # more than 1 / 16 of the nodes deleted: HBM frontier again
    more = [int(v) for v in rng.permutation(n) if v not in set(dead.tolist())][: n // 12]
    for lab in more:
        assert g.remove(lab) == 0
    g.flush()
    o2 = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    assert _check(g, o2, Q, 10, 64).last_visited_mode == 0
