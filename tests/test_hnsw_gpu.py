"""GPU parity tests for the HNSW path through the C ABI against the CPU oracle.

Same inputs, single-threaded inserts => the product's host graph equals the oracle's graph
(tests/test_host_graph.py), so the device search must return the oracle's neighbours: identical ids
and distance bits on data without exact distance ties; recall >= the oracle's where ties exist or
the build is multi-threaded (the reference's own bar: recall, vector_test.cc:439-500)."""
import ctypes as C

import numpy as np
import pytest

from conftest import reference_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _pair(vsa, oracle, x, metric, M=16, efc=100, ef=10, labels=None, cap=None, threads=1):
    n, dim = x.shape
    cap = cap or n
    g = vsa.Index("HNSW", dim, metric, initial_cap=cap, m=M, ef_construction=efc, ef_runtime=ef, build_threads=threads)
    o = oracle.HNSW(dim, metric, max_elements=cap, M=M, ef_construction=efc, ef=ef)
    if threads == 1:
        for i in range(n):
            lab = int(labels[i]) if labels is not None else i
            assert g.add(lab, x[i]) == 0
    else:
        g.add_batch(x, labels)
    o.add_many(x, labels)
    return g, o


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("metric", ["L2", "IP", "COSINE"])
@pytest.mark.parametrize("n,dim,M", [(3000, 64, 16), (2000, 100, 8), (1500, 768, 16)])
def test_search_equals_oracle(vsa, oracle, metric, n, dim, M):
    rng = np.random.default_rng(21)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == "COSINE":
        x = np.stack([oracle.normalize(v)[0] for v in x])
    g, o = _pair(vsa, oracle, x, metric, M=M)
    st = g.stats()
    assert st.count == n and st.max_level == o.max_level and st.entry_point == o.entry_point
    Q = rng.standard_normal((12, dim)).astype(np.float32)
    if metric == "COSINE":
        Q = np.stack([oracle.normalize(v)[0] for v in Q])
    for ef in (0, 10, 64, 128, 300):
        for q in Q[:4]:
            _same(*g.search(q, 10, ef=ef), *o.search(q, 10, ef=ef))
    # batched: same answers, and the layer-0 work counters match the oracle's
    D, L, N = g.search_batch(Q, 10, ef=128)
    ne = nh = 0
    for i in range(len(Q)):
        od, ol, e, h = o.search(Q[i], 10, ef=128, stats=True)
        _same(D[i, :N[i]], L[i, :N[i]], od, ol)
        ne += e
        nh += h
    st = g.stats()
    assert (st.last_n_eval, st.last_n_hops) == (ne, nh)


def test_known_answer_cosine_scores_hnsw(vsa, oracle):
    dim = 100
    g = vsa.Index("HNSW", dim, "COSINE", initial_cap=200)
    for d in range(100):
        v = np.zeros(dim, np.float32)
        v[0], v[1] = 1, d
        assert g.add(d, oracle.normalize(v)[0]) == 0
    q = np.zeros(dim, np.float32)
    q[0] = 1
    d, l = g.search(oracle.normalize(q)[0], 3, ef=1)
    assert l.tolist() == [0, 1, 2]
    assert ["%.12g" % v for v in d] == ["0", "0.292893230915", "0.552786409855"]


def test_ef_runtime_recall_like_vector_test(vsa, oracle):
    """vector_test.cc:439-500: 1000x100 deterministic vectors, M=16 efC=20: recall@10 vs FLAT >= 0.96."""
    x = reference_vectors(1000, 100, 2.2)
    g = vsa.Index("HNSW", 100, "L2", initial_cap=31000, m=16, ef_construction=20, ef_runtime=20)
    f = vsa.Index("FLAT", 100, "L2", initial_cap=31000)
    for i in range(1000):
        assert g.add(i, x[i]) == 0
    f.add_batch(x)
    Q = reference_vectors(50, 100, 1.5)

    def recall(ef):
        _, Lh, _ = g.search_batch(Q, 10, ef=ef)
        _, Lf, _ = f.search_batch(Q, 10)
        return sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Lh, Lf)) / 500.0

    assert recall(160) >= 0.96
    assert recall(20) == recall(0)


def test_filter_and_tombstones(vsa, oracle):
    rng = np.random.default_rng(22)
    n, dim = 4000, 48
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, "L2", M=12, efc=80)
    allowed = np.sort(rng.choice(n, n // 10, replace=False))
    bits = oracle.allow_bitmap(allowed, n)
    Q = rng.standard_normal((6, dim)).astype(np.float32)
    for q in Q:
        _same(*g.search(q, 10, ef=64, allow=bits, allow_nbits=n), *o.search(q, 10, ef=64, allow=bits, allow_nbits=n))
    for lab in range(0, 400):
        assert g.remove(lab) == 0 and o.mark_delete(lab) == 0
    assert g.remove(5) != 0                      # already deleted
    assert g.stats().deleted == 400
    for q in Q:
        gd, gl = g.search(q, 10, ef=64)
        _same(gd, gl, *o.search(q, 10, ef=64))
        assert gl.min() >= 400
        _same(*g.search(q, 10, ef=64, allow=bits, allow_nbits=n), *o.search(q, 10, ef=64, allow=bits, allow_nbits=n))
    assert g.distance(5, Q[0]) is None and not g.contains(5)
    assert g.distance(1000, Q[0]).view(np.uint32) == o.distance(1000, Q[0]).view(np.uint32)


@pytest.mark.parametrize("selectivity", [0.3, 0.05])
def test_filtered_search_keeps_the_whole_frontier(vsa, oracle, selectivity):
    """With a selective filter (and tombstones) the result list fills slowly, so hnswlib's candidate_set grows to
    about ef / selectivity entries before anything can be pruned -- thousands, far beyond the LDS pool.  The
    device keeps that frontier in HBM; dropping entries (what a fixed pool did) ends the search early and misses
    closer allowed nodes.  ids, distance bits and the eval / hop counts must be the oracle's."""
    rng = np.random.default_rng(222)
    n, dim = 6000, 96
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, "IP", M=8, efc=40)
    for lab in rng.choice(n, n // 8, replace=False):
        assert g.remove(int(lab)) == 0 and o.mark_delete(int(lab)) == 0
    bits = oracle.allow_bitmap(np.flatnonzero(rng.random(n) < selectivity), n)
    Q = rng.standard_normal((24, dim)).astype(np.float32)
    D, L, N = g.search_batch(Q, 50, ef=256, allow=bits, allow_nbits=n)
    ne = nh = 0
    for i in range(len(Q)):
        od, ol, e, h = o.search(Q[i], 50, ef=256, allow=bits, allow_nbits=n, stats=True)
        _same(D[i, :N[i]], L[i, :N[i]], od, ol)
        ne += e
        nh += h
    st = g.stats()
    assert (st.last_n_eval, st.last_n_hops) == (ne, nh)
    _same(*g.search(Q[0], 50, ef=256, allow=bits, allow_nbits=n), *o.search(Q[0], 50, ef=256, allow=bits, allow_nbits=n))


def test_search_test_cases_hnsw(vsa, oracle):
    """search_test.cc:793-899 with HNSW(M=10, efC=300, ef=30) on 10000 collinear-ish vectors."""
    N = 10000
    x = reference_vectors(N, 100, 10.0)
    g = vsa.Index("HNSW", 100, "L2", initial_cap=N, m=10, ef_construction=300, ef_runtime=30, build_threads=8)
    g.add_batch(x)
    q = np.zeros(100, np.float32)
    d, l = g.search(q, 5, ef=30)
    assert set(l.tolist()) == {0, 1, 2, 3, 4}
    cases = [(lambda i: i < 5, {0, 1, 2, 3, 4}), (lambda i: i < 3, {0, 1, 2}), (lambda i: not (0 <= i <= 100), {101, 102, 103, 104, 105}),
             (lambda i: not i < 5, {5, 6, 7, 8, 9}), (lambda i: 4 <= i <= 100 and i < 5, {4}), (lambda i: False, set())]
    for pred, expected in cases:
        allowed = [i for i in range(N) if pred(i)]
        if len(allowed) <= 0.001 * N:       # planner.cc:21-45 -> pre-filter path
            d, l = g.search_labels(q, 5, np.array(allowed, np.uint64))
        else:
            d, l = g.search(q, 5, ef=30, allow=oracle.allow_bitmap(allowed, N), allow_nbits=N)
        assert set(l.tolist()) == expected


def test_parallel_build_recall(vsa, oracle):
    rng = np.random.default_rng(23)
    n, dim = 20000, 64
    A = rng.standard_normal((dim, 16)).astype(np.float32)
    x = (rng.standard_normal((n, 16)).astype(np.float32) @ A.T + 0.05 * rng.standard_normal((n, dim)).astype(np.float32))
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, build_threads=8)
    g.add_batch(x)
    f = vsa.Index("FLAT", dim, "L2", initial_cap=n)
    f.add_batch(x)
    Q = (rng.standard_normal((64, 16)).astype(np.float32) @ A.T).astype(np.float32)
    _, Lh, _ = g.search_batch(Q, 10, ef=128)
    _, Lf, _ = f.search_batch(Q, 10)
    rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Lh, Lf)) / 640.0
    assert rec >= 0.95, rec


def test_resize_modify_and_cancel(vsa, oracle):
    rng = np.random.default_rng(24)
    dim = 32
    x = rng.standard_normal((300, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x[:200], "L2", M=8, efc=50, cap=200)
    assert g.add(200, x[200]) == vsa.VK_ERR_CAPACITY
    g.resize(300)
    o.resize(300)
    for i in range(200, 300):
        assert g.add(i, x[i]) == 0 and o.add(x[i], i) == 0
    assert g.stats().capacity == 300
    q = rng.standard_normal(dim).astype(np.float32)
    _same(*g.search(q, 10, ef=50), *o.search(q, 10, ef=50))
    row = rng.standard_normal(dim).astype(np.float32)
    assert g.add(17, row) == 0                      # modify in place (updatePoint)
    assert np.array_equal(g.get_row(17), row)
    d, l = g.search(row, 1, ef=50)
    assert l.tolist() == [17] and d[0] == 0
    flag = C.c_int(1)
    with pytest.raises(vsa.VkError) as e:
        g.search(q, 5, cancel=flag, partial_ok=False)
    assert e.value.code == vsa.VK_ERR_CANCELLED


def test_save_load_round_trip_and_validation(vsa, oracle):
    rng = np.random.default_rng(25)
    n, dim, M = 1500, 40, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, "L2", M=M, efc=60)
    for lab in (3, 4, 5):
        g.remove(lab)
        o.mark_delete(lab)
    chunks = g.save()
    sl0 = 2 * M * 4 + 4
    assert len(chunks[1]) == sl0 + dim * 4 + 8        # [level-0 links | vector | label]
    e = o.export_graph()
    for i in (0, 7, 1499):                            # the chunk stream carries hnswlib's level-0 record
        words = np.frombuffer(chunks[1 + i][:sl0], dtype=np.uint32)
        assert (words[0] & 0xFFFF) == e["l0"][i, 0] and words[1:1 + e["l0"][i, 0]].tolist() == e["l0"][i, 1:1 + e["l0"][i, 0]].tolist()
        assert bool(words[0] & 0x10000) == bool(e["deleted"][i])
    g2 = vsa.Index.load(chunks, "HNSW", dim, "L2", m=M, ef_construction=60)
    assert g2.stats().count == n and g2.stats().deleted == 3
    q = rng.standard_normal(dim).astype(np.float32)
    _same(*g2.search(q, 10, ef=80), *o.search(q, 10, ef=80))
    # loadCheck-style rejections (hnswalg.h:872-885, vector_test.cc:1004-1203)
    with pytest.raises(vsa.VkError):
        vsa.Index.load(chunks, "HNSW", dim, "L2", m=M + 1, ef_construction=60)     # M mismatch
    bad = list(chunks)
    rec = bytearray(bad[5])
    rec[4:8] = (10 ** 6).to_bytes(4, "little")                                      # neighbour id out of range
    bad[5] = bytes(rec)
    with pytest.raises(vsa.VkError) as ei:
        vsa.Index.load(bad, "HNSW", dim, "L2", m=M, ef_construction=60)
    assert "neighbor id out of range" in ei.value.msg
    with pytest.raises(vsa.VkError):
        vsa.Index.load(chunks[:100], "HNSW", dim, "L2", m=M, ef_construction=60)    # truncated stream


def test_multithreaded_build_same_graph_same_answers(vsa, oracle):
    """Build with 8 host threads (graph differs from any single-threaded build), hand the SAME graph to the
    oracle through the SaveIndex chunk stream, and require identical neighbours from GPU and CPU."""
    rng = np.random.default_rng(26)
    n, dim, M = 12000, 96, 16
    A = rng.standard_normal((dim, 24)).astype(np.float32)
    x = (rng.standard_normal((n, 24)).astype(np.float32) @ A.T + 0.1 * rng.standard_normal((n, dim)).astype(np.float32))
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=100, build_threads=8)
    g.add_batch(x)
    o = oracle.HNSW.from_saved_chunks(g.save(), dim, "L2", M, ef_construction=100)
    assert o.count == n and o.max_level == g.stats().max_level and o.entry_point == g.stats().entry_point
    Q = (rng.standard_normal((40, 24)).astype(np.float32) @ A.T).astype(np.float32)
    D, L, N = g.search_batch(Q, 10, ef=128)
    for i in range(len(Q)):
        _same(D[i, :N[i]], L[i, :N[i]], *o.search(Q[i], 10, ef=128))


@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_large_ef_uses_the_lds_result_list(vsa, oracle, metric):
    """512 < ef <= 4096: the result list moves from the lanes' registers to LDS (one wave per block); the
    answers must still be the oracle's on the same graph -- including k > 512 and a filter."""
    rng = np.random.default_rng(61)
    n, dim = 6000, 40
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, metric, M=12, efc=80)
    Q = rng.standard_normal((12, dim)).astype(np.float32)
    for q in Q:
        for k, ef in ((10, 600), (10, 1500), (700, 1000), (2000, 4096)):
            _same(*g.search(q, k, ef=ef), *o.search(q, k, ef=ef))
    allowed = np.sort(rng.choice(n, 2500, replace=False)).astype(np.uint64)
    bits = oracle.allow_bitmap(allowed, n)
    for q in Q[:4]:
        _same(*g.search(q, 50, ef=900, allow=bits, allow_nbits=n), *o.search(q, 50, ef=900, allow=bits, allow_nbits=n))
    # batched, and the documented ceiling
    D, L, N = g.search_batch(Q, 20, ef=800)
    for i, q in enumerate(Q):
        _same(D[i, :N[i]], L[i, :N[i]], *o.search(q, 20, ef=800))
    _same(*g.search(Q[0], 10, ef=5000), *o.search(Q[0], 10, ef=5000))      # (the ceiling is 16384 now)
    with pytest.raises(vsa.VkError):
        g.search(Q[0], 10, ef=20000)


def test_long_rows(vsa, oracle):
    """D = 1536 (6 KB rows): the LDS query block and the 16-rows-per-round gathers at a long row length."""
    rng = np.random.default_rng(71)
    n, dim = 1200, 1536
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, "COSINE" if False else "IP", M=16, efc=60)
    for q in rng.standard_normal((6, dim)).astype(np.float32):
        _same(*g.search(q, 10, ef=80), *o.search(q, 10, ef=80))


def test_k_up_to_the_default_max_vector_knn_and_wide_graphs(vsa, oracle):
    """ft_search_parser.cc:34-45: max-vector-knn defaults to 10000 -- k (hence ef) that large must be served, and so must
    graphs wider than M = 128"""
    rng = np.random.default_rng(31)
    n, dim = 12_000, 32
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g, o = _pair(vsa, oracle, x, "L2", M=8, efc=40, threads=1)
    Q = rng.standard_normal((3, dim)).astype(np.float32)
    for k, ef in ((5000, 0), (10000, 0), (10, 6000)):
        D, L, N = g.search_batch(Q, k, ef=ef)
        for i in range(len(Q)):
            od, ol = o.search(Q[i], k, ef=ef)
            assert N[i] == len(ol)
            _same(D[i, :N[i]], L[i, :N[i]], od, ol)
    with pytest.raises(vsa.VkError):                     # beyond the LDS list: a clean error, not a wrong answer
        g.search_batch(Q, 20000)
    w, ow = _pair(vsa, oracle, x[:3000], "L2", M=150, efc=320, threads=1)
    for q in Q:
        _same(*w.search(q, 10, ef=64), *ow.search(q, 10, ef=64))
