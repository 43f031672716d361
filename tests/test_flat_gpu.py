"""GPU parity tests for the FLAT path, through the C ABI (include/vk_index.h) against the
CPU oracle (oracle/).  Bar: bit-exact neighbour ids AND distance bits, ties included."""
import ctypes as C

import numpy as np
import pytest

from conftest import reference_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _data(n, dim, seed, unit=False):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x


def _prep(oracle, x, metric):
    if metric == "COSINE":
        return np.stack([oracle.normalize(v)[0] for v in x])
    return x


def _both(vsa, oracle, x, metric, labels=None, cap=None):
    n, dim = x.shape
    cap = cap or n
    g = vsa.Index("FLAT", dim, metric, initial_cap=cap)
    g.add_batch(x, labels)
    o = oracle.Flat(dim, metric, max_elements=cap)
    o.add_many(x, labels)
    return g, o


def _assert_same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("metric", ["L2", "IP", "COSINE"])
@pytest.mark.parametrize("n,dim", [(20000, 128), (5000, 100), (6000, 768), (300, 7), (1000, 16), (257, 1)])
def test_single_query_bit_exact(vsa, oracle, metric, n, dim):
    x = _prep(oracle, _data(n, dim, 1), metric)
    g, o = _both(vsa, oracle, x, metric)
    qs = _prep(oracle, _data(8, dim, 2), metric)
    for q in qs:
        gd, gl = g.search(q, 10)
        od, ol = o.search(q, 10)
        _assert_same(gd, gl, od, ol)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8, 17, 64])
def test_batch_queries_bit_exact(vsa, oracle, nq):
    x = _data(30000, 128, 3)
    g, o = _both(vsa, oracle, x, "L2")
    Q = _data(nq, 128, 4)
    D, L, N = g.search_batch(Q, 10)
    for i in range(nq):
        od, ol = o.search(Q[i], 10)
        assert N[i] == 10
        _assert_same(D[i], L[i], od, ol)


@pytest.mark.parametrize("k", [1, 10, 64, 65, 100, 256, 300, 1000, 1024, 1025, 2500, 5000])
def test_k_values(vsa, oracle, k):
    x = _data(5000, 64, 5)
    g, o = _both(vsa, oracle, x, "L2")
    q = _data(1, 64, 6)[0]
    gd, gl = g.search(q, k)
    od, ol = o.search(q, k)
    _assert_same(gd, gl, od, ol)


def test_k_clamped_to_count_and_empty(vsa, oracle):
    x = _data(7, 32, 7)
    g, o = _both(vsa, oracle, x, "IP", cap=100)
    q = _data(1, 32, 8)[0]
    gd, gl = g.search(q, 10)
    od, ol = o.search(q, 10)
    assert len(gl) == 7
    _assert_same(gd, gl, od, ol)
    e = vsa.Index("FLAT", 32, "L2", initial_cap=10)
    d, l = e.search(q, 5)
    assert len(l) == 0


def test_ties_resolve_by_label(vsa, oracle):
    """duplicate rows with different labels: the k smallest by (distance,label) (bruteforce.h heap)."""
    base = _data(50, 48, 9)
    x = np.concatenate([base, base, base])  # each row three times
    labels = np.random.default_rng(10).permutation(1000)[:150].astype(np.uint64)
    g, o = _both(vsa, oracle, x, "L2", labels=labels)
    for q in base[:5]:
        gd, gl = g.search(q, 4)
        od, ol = o.search(q, 4)
        assert gd[0] == 0 and gd[2] == 0
        _assert_same(gd, gl, od, ol)


@pytest.mark.parametrize("copies", [6000, 700])
@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_massive_distance_ties_are_broken_by_label(vsa, oracle, copies, metric):
    """A few distinct rows, each stored thousands of times under shuffled labels: every partial list the scan leaves
    is full of equal distances, so the merge has to pick the k smallest LABELS at the k-th distance (the selection
    merge's second descent; with 6000 copies the survivors overflow its fast path as well).  Single query, a batch
    on the scan kernel and a batch on the matrix-core path."""
    dim = 64
    base = _data(3, dim, 91)
    x = np.repeat(base, copies, axis=0)
    labels = (np.random.default_rng(92).permutation(len(x)).astype(np.uint64) + 7)
    g, o = _both(vsa, oracle, x, metric, labels=labels)
    Q = np.concatenate([base, _data(5, dim, 93)])
    for k in (1, 10, 64):
        for q in Q[:4]:
            _assert_same(*g.search(q, k), *o.search(q, k))
        D, L, N = g.search_batch(Q, k)
        for i in range(len(Q)):
            od, ol = o.search(Q[i], k)
            assert N[i] == k
            _assert_same(D[i], L[i], od, ol)


def test_reference_generator_collinear_data(vsa, oracle):
    """testing/common.cc vectors: near-collinear rows, many near ties (search_test.cc fixture)."""
    x = reference_vectors(10000, 100, 10.0)
    g, o = _both(vsa, oracle, x, "L2")
    for q in (np.zeros(100, np.float32), np.ones(100, np.float32), x[4321]):
        gd, gl = g.search(q, 10)
        od, ol = o.search(q, 10)
        _assert_same(gd, gl, od, ol)
    gd, gl = g.search(np.zeros(100, np.float32), 5)
    assert gl.tolist() == [0, 1, 2, 3, 4]


def test_known_answer_cosine_scores_through_abi(vsa, oracle):
    """vector_search_integration_test.py:143-165 score strings, FLAT."""
    dim = 100
    g = vsa.Index("FLAT", dim, "COSINE", initial_cap=200)
    for d in range(100):
        v = np.zeros(dim, np.float32)
        v[0], v[1] = 1, d
        assert g.add(d, oracle.normalize(v)[0]) == 0
    q = np.zeros(dim, np.float32)
    q[0] = 1
    d, l = g.search(oracle.normalize(q)[0], 3)
    assert l.tolist() == [0, 1, 2]
    assert ["%.12g" % v for v in d] == ["0", "0.292893230915", "0.552786409855"]


def test_filter_bitmap(vsa, oracle):
    x = _data(20000, 64, 11)
    g, o = _both(vsa, oracle, x, "L2")
    rng = np.random.default_rng(12)
    allowed = np.sort(rng.choice(20000, 2000, replace=False))
    bits = oracle.allow_bitmap(allowed, 20000)
    q = _data(1, 64, 13)[0]
    gd, gl = g.search(q, 10, allow=bits, allow_nbits=20000)
    # exact k best among the allowed rows == pre-filter answer on distinct distances
    od, ol = oracle.prefilter_topk("L2", q, x[allowed], allowed.astype(np.uint64), 10)
    _assert_same(gd, gl, od, ol)
    # fewer allowed than k
    bits2 = oracle.allow_bitmap([5, 77, 1234], 20000)
    gd, gl = g.search(q, 10, allow=bits2, allow_nbits=20000)
    assert sorted(gl.tolist()) == [5, 77, 1234] and np.all(np.diff(gd) >= 0)
    # labels beyond allow_nbits are rejected
    gd, gl = g.search(q, 10, allow=oracle.allow_bitmap(range(64), 64), allow_nbits=64)
    assert max(gl.tolist()) < 64 and len(gl) == 10


def test_mutation_sequence_matches_oracle(vsa, oracle):
    """addPoint overwrite, removePoint swap-delete, capacity error, resize (bruteforce.h:66-113)."""
    rng = np.random.default_rng(14)
    dim, cap = 24, 300
    g = vsa.Index("FLAT", dim, "L2", initial_cap=cap)
    o = oracle.Flat(dim, "L2", max_elements=cap)
    live = set()
    nxt = 0
    for step in range(1500):
        r = rng.random()
        if r < 0.55 or not live:
            row = rng.standard_normal(dim).astype(np.float32)
            rc_g, rc_o = g.add(nxt, row), o.add(row, nxt)
            assert (rc_g == vsa.VK_ERR_CAPACITY) == (rc_o == 1)
            if rc_o == 0:
                live.add(nxt)
            nxt += 1
        elif r < 0.8:
            lab = int(rng.choice(sorted(live)))
            g.remove(lab)
            o.remove(lab)
            live.discard(lab)
        elif r < 0.9:
            lab = int(rng.choice(sorted(live)))     # modify == add with a known label
            row = rng.standard_normal(dim).astype(np.float32)
            assert g.add(lab, row) == 0 and o.add(row, lab) == 0
        else:
            q = rng.standard_normal(dim).astype(np.float32)
            gd, gl = g.search(q, 7)
            od, ol = o.search(q, 7)
            _assert_same(gd, gl, od, ol)
        if step == 900:
            g.resize(cap + 200)
            o.resize(cap + 200)
    assert g.stats().count == o.count == len(live)
    assert g.stats().capacity == 500
    g.remove(10 ** 9)  # unknown label: silently ignored
    for lab in list(live)[:20]:
        assert g.contains(lab)
        row = g.get_row(lab)
        q = rng.standard_normal(dim).astype(np.float32)
        assert g.distance(lab, q).view(np.uint32) == o.distance(lab, q).view(np.uint32)
        assert row is not None
    assert g.distance(10 ** 9, np.zeros(dim, np.float32)) is None


def test_search_labels_prefilter_rule(vsa, oracle):
    """vector_base.cc:509-530: strict `<` replacement, ties keep the earlier key."""
    base = _data(200, 32, 15)
    x = np.concatenate([base, base])        # ties between label i and i+200
    g, o = _both(vsa, oracle, x, "L2")
    rng = np.random.default_rng(16)
    for _ in range(5):
        labels = rng.permutation(400)[:150].astype(np.uint64)
        q = base[int(rng.integers(200))]
        gd, gl = g.search_labels(q, 5, labels)
        od, ol = oracle.prefilter_topk("L2", q, x[labels.astype(int)], labels, 5)
        _assert_same(gd, gl, od, ol)
    # unknown labels are skipped
    gd, gl = g.search_labels(base[0], 3, np.array([10 ** 6, 0, 200, 5], np.uint64))
    assert set(gl.tolist()) <= {0, 200, 5} and len(gl) == 3


def test_cancel_before_start_looks_at_first_k_rows_only(vsa, oracle):
    x = _data(5000, 32, 17)
    g, o = _both(vsa, oracle, x, "L2")
    q = _data(1, 32, 18)[0]
    flag = C.c_int(1)
    gd, gl = g.search(q, 10, cancel=flag)
    od, ol = o.search(q, 10, cancel_after=0)
    _assert_same(gd, gl, od, ol)
    flag = C.c_int(0)
    gd, gl = g.search(q, 10, cancel=flag)
    od, ol = o.search(q, 10)
    _assert_same(gd, gl, od, ol)


def test_save_load_round_trip(vsa, oracle):
    """bruteforce.h:147-207 chunk stream: header proto + one chunk per element."""
    x = _data(1234, 40, 19)
    labels = (np.arange(1234) * 3 + 7).astype(np.uint64)
    g, o = _both(vsa, oracle, x, "L2", labels=labels, cap=2000)
    chunks = g.save()
    assert len(chunks) == 1 + 1234 and len(chunks[1]) == 40 * 4 + 8
    # header = BruteForceIndexHeader{1: max_elements, 2: size_per_element, 3: curr_element_count}
    assert chunks[0] == bytes([0x08, 0xD0, 0x0F, 0x10, 0xA8, 0x01, 0x18, 0xD2, 0x09])
    g2 = vsa.Index.load(chunks, "FLAT", 40, "L2")
    assert g2.stats().count == 1234 and g2.stats().capacity == 2000
    q = _data(1, 40, 20)[0]
    _assert_same(*g2.search(q, 10), *o.search(q, 10))


@pytest.mark.parametrize("metric", ["IP", "COSINE"])
@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 256, 10), (5000, 100, 33, 10), (9000, 768, 64, 10), (4097, 16, 16, 5),
                                        (3000, 48, 100, 64), (300, 7, 40, 10),
                                        (3000, 1024, 50, 10), (2500, 1536, 40, 10), (2000, 1000, 70, 10)])   # 24 / 16 queries per tile
def test_mfma_batched_path_bit_exact(vsa, oracle, metric, n, dim, nq, k):
    """K4 (flat_gemm.hip): >= 5 queries in the inner-product space go through the f32 MFMA kernel; one
    accumulator tile per SimSIMD lane class keeps the result bit-identical to the CPU reference."""
    x = _prep(oracle, _data(n, dim, 31), metric)
    g, o = _both(vsa, oracle, x, metric)
    Q = _prep(oracle, _data(nq, dim, 32), metric)
    D, L, N = g.search_batch(Q, k)
    for i in range(nq):
        od, ol = o.search(Q[i], k)
        assert N[i] == len(ol)
        _assert_same(D[i, :N[i]], L[i, :N[i]], od, ol)


def test_mfma_path_ties_and_filter(vsa, oracle):
    base = _data(400, 64, 33)
    x = np.concatenate([base, base, base])
    labels = np.random.default_rng(34).permutation(5000)[:1200].astype(np.uint64)
    g, o = _both(vsa, oracle, x, "IP", labels=labels)
    Q = base[:32]
    D, L, N = g.search_batch(Q, 6)
    for i in range(32):
        _assert_same(D[i, :N[i]], L[i, :N[i]], *o.search(Q[i], 6))
    # with a filter the batched answer must equal the single-query scan answer (K3), itself checked above
    allowed = np.sort(np.random.default_rng(35).choice(labels, 300, replace=False))
    bits = oracle.allow_bitmap(allowed, 5000)
    D, L, N = g.search_batch(Q, 6, allow=bits, allow_nbits=5000)
    for i in range(32):
        sd, sl = g.search(Q[i], 6, allow=bits, allow_nbits=5000)
        _assert_same(D[i, :N[i]], L[i, :N[i]], sd, sl)
        assert set(L[i, :N[i]].tolist()) <= set(allowed.tolist())


def test_large_k_pages_through_ties_and_filter(vsa, oracle):
    """k > 1024 is served in passes bounded by the previous pass's last (distance,label): duplicates that
    straddle a pass boundary must not be lost or repeated."""
    base = _data(900, 32, 51)
    x = np.concatenate([base, base, base])          # every distance occurs three times
    g, o = _both(vsa, oracle, x, "L2")
    q = _data(1, 32, 52)[0]
    gd, gl = g.search(q, 2600)
    od, ol = o.search(q, 2600)
    _assert_same(gd, gl, od, ol)
    D, L, N = g.search_batch(_data(3, 32, 53), 1500)
    for i in range(3):
        od, ol = o.search(_data(3, 32, 53)[i], 1500)
        _assert_same(D[i, :N[i]], L[i, :N[i]], od, ol)


def test_baseline_config1_shape(vsa, oracle):
    """BASELINE.json configs[0]: FLAT 100k x 128 f32 L2 k=10, one query per call -- ids and distance
    bits of every answer against the oracle (the reference's own CPU-runnable case)."""
    n, dim, k = 100_000, 128, 10
    x = _data(n, dim, 1234)
    g, o = _both(vsa, oracle, x, "L2")
    Q = _data(64, dim, 1235)
    for q in Q:
        _assert_same(*g.search(q, k), *o.search(q, k))
    # the same queries as one batch
    D, L, Nn = g.search_batch(Q, k)
    for i, q in enumerate(Q):
        _assert_same(D[i, :Nn[i]], L[i, :Nn[i]], *o.search(q, k))


@pytest.mark.parametrize("k", [11, 64, 100, 256, 300])
def test_mfma_path_larger_k(vsa, oracle, k):
    """k > 10 keeps the per-lane lists in HBM scratch (k <= 256 on the matrix-core path, beyond that the
    scan kernel): same answers either way."""
    n, dim, nq = 12000, 64, 24
    x = _prep(oracle, _data(n, dim, 51), "COSINE")
    g, o = _both(vsa, oracle, x, "COSINE")
    Q = _prep(oracle, _data(nq, dim, 52), "COSINE")
    D, L, N = g.search_batch(Q, k)
    for i in range(nq):
        od, ol = o.search(Q[i], k)
        _assert_same(D[i, :N[i]], L[i, :N[i]], od, ol)
