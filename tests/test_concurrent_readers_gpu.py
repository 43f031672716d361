"""Many reader threads on one index, the way the module's reader pool calls Search (search.cc:886-910): single
queries, batches of every kernel class, filtered queries and key-list queries in flight at once, with and without
the coalescer.  Every answer must be the answer the same call gives alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
@pytest.mark.parametrize("coalesce", [False, True])
def test_mixed_calls_from_many_threads(vsa, oracle, algo, coalesce):
    rng = np.random.default_rng(77)
    n, dim = (20000, 128) if algo == "FLAT" else (6000, 64)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    kw = dict(m=16, ef_construction=100, ef_runtime=64) if algo == "HNSW" else {}
    g = vsa.Index(algo, dim, "L2", initial_cap=n, **kw)
    g.add_batch(x)
    bits = oracle.allow_bitmap(np.arange(0, n, 3, dtype=np.uint64), n)
    keys = np.arange(5, n, 37, dtype=np.uint64)
    jobs = []
    for i in range(96):
        kind = ("one", "batch2", "batch9", "batch40", "filtered", "keys")[i % 6]
        q = rng.standard_normal((40, dim)).astype(np.float32)
        jobs.append((kind, q))

    def call(kind, q):
        if kind == "one":
            return g.search(q[0], 10)
        if kind == "batch2":
            return g.search_batch(q[:2], 10)[:2]
        if kind == "batch9":
            return g.search_batch(q[:9], 10)[:2]
        if kind == "batch40":
            return g.search_batch(q, 10)[:2]
        if kind == "filtered":
            return g.search(q[0], 10, allow=bits, allow_nbits=n)
        return g.search_labels(q[0], 10, keys)

    want = [call(k, q) for k, q in jobs]            # one at a time first
    if coalesce:
        g.set_coalescing(32, 200)
    got = [None] * len(jobs)
    errs = []

    def worker(t):
        try:
            for j in range(t, len(jobs), 16):
                for _ in range(3):
                    got[j] = call(*jobs[j])
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    g.set_coalescing(0, 0)
    assert not errs, errs[:3]
    for (kind, _), a, b in zip(jobs, want, got):
        assert np.asarray(a[1]).tolist() == np.asarray(b[1]).tolist(), kind
        assert np.asarray(a[0]).view(np.uint32).tolist() == np.asarray(b[0]).view(np.uint32).tolist(), kind
