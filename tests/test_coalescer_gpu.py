"""N1 -- query coalescing: concurrent single-query vk_index_search calls (the reference's
reader-pool pattern, search.cc:886-910) merged into device batches must each get exactly the
answer they would have got alone (= the oracle's)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _run_threads(ix, Q, k, ef, nthreads):
    out = [None] * len(Q)
    err = []
    nxt = [0]
    lock = threading.Lock()

    def worker():
        try:
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(Q):
                    return
                out[i] = ix.search_one(Q[i], k, ef)
        except Exception as e:  # pragma: no cover
            err.append(e)

    ts = [threading.Thread(target=worker) for _ in range(nthreads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    return out


@pytest.mark.parametrize("metric", ["IP", "L2"])
def test_flat_coalesced_answers_equal_oracle(vsa, oracle, metric):
    rng = np.random.default_rng(11)
    n, dim, k = 20000, 96, 10
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((300, dim)).astype(np.float32)
    ix = vsa.Index("FLAT", dim, metric, initial_cap=n)
    ix.add_batch(X)
    ref = oracle.Flat(dim, metric, max_elements=n)
    ref.add_many(X)
    ix.set_coalescing(32, 2000)
    got = _run_threads(ix, Q, k, 0, 48)
    for i in range(len(Q)):
        d, l = ref.search(Q[i], k)
        assert np.array_equal(got[i][1], l), i
        assert np.array_equal(got[i][0].view(np.uint32), d.view(np.uint32)), i
    st = ix.stats()
    assert st.coalesced_queries == len(Q)
    assert st.coalesced_batches < len(Q)          # at least some calls shared a launch
    # turning it off returns to one launch per call, same answers
    ix.set_coalescing(0, 0)
    d, l = ix.search_one(Q[0], k)
    assert np.array_equal(l, got[0][1])
    assert ix.stats().coalesced_queries == len(Q)


def test_hnsw_coalesced_lanes_by_k_and_ef(vsa):
    rng = np.random.default_rng(12)
    n, dim = 6000, 64
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((128, dim)).astype(np.float32)
    ix = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, ef_runtime=64)
    ix.add_batch(X)
    alone = {(k, ef): [ix.search_one(q, k, ef) for q in Q] for (k, ef) in ((5, 64), (10, 128))}
    ix.set_coalescing(16, 1000)
    res = {}

    def lane(k, ef):
        res[(k, ef)] = _run_threads(ix, Q, k, ef, 24)

    ts = [threading.Thread(target=lane, args=ke) for ke in alone]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for ke in alone:
        for i in range(len(Q)):
            assert np.array_equal(res[ke][i][1], alone[ke][i][1]), (ke, i)
            assert np.array_equal(res[ke][i][0], alone[ke][i][0]), (ke, i)
    st = ix.stats()
    assert st.coalesced_queries == 2 * len(Q) and st.coalesced_batches < 2 * len(Q)


def test_single_caller_is_not_stalled_beyond_the_wait(vsa):
    import time
    rng = np.random.default_rng(13)
    X = rng.standard_normal((1000, 32)).astype(np.float32)
    ix = vsa.Index("FLAT", 32, "L2", initial_cap=1000)
    ix.add_batch(X)
    ix.search_one(X[0], 3)
    ix.set_coalescing(64, 20000)      # 20 ms window, nobody else arrives
    t0 = time.perf_counter()
    d, l = ix.search_one(X[5], 3)
    dt = time.perf_counter() - t0
    assert l[0] == 5 and d[0] == 0.0
    assert 0.015 < dt < 1.0
