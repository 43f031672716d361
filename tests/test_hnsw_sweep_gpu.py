"""Randomised parameter sweep over the HNSW search path: size, dimension, M, efConstruction, efSearch, k, batch
size, metric, row storage, filter and tombstones drawn from a fixed seed.  The host builds the graph single-threaded
(link for link the oracle's graph), so the device search must return the oracle's ids and distance bits: small
batches run the latency kernel (split rounds, 48 pieces in flight), more than 512 queries the throughput kernel,
ef > 512 the LDS result list."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# VK_SWEEP_OFFSET=<n> shifts every seed: a different set of shapes for a one-off hunt
SWEEP_OFFSET = int(__import__("os").environ.get("VK_SWEEP_OFFSET", "0"))


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _bf16_round(x):
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("seed", range(48))
def test_random_shape(vsa, oracle, seed):
    rng = np.random.default_rng(9000 + seed + SWEEP_OFFSET)
    dim = int(rng.choice([1, 7, 16, 48, 100, 128, 200, 384, 768, 1000, 1536]))
    n = int(rng.integers(50, 2500 if dim <= 200 else 900))
    M = int(rng.choice([4, 8, 16, 16, 32, 48]))
    efc = int(rng.choice([20, 100, 200]))
    ef = int(rng.choice([0, 10, 64, 128, 128, 200, 256, 300, 512, 600, 1000]))
    k = int(min(n, rng.choice([1, 5, 10, 10, 50, 100])))
    nq = int(rng.choice([1, 3, 16, 64, 600]))
    metric = str(rng.choice(["L2", "IP", "COSINE"]))
    dtype = "bf16" if rng.random() < 0.25 else "f32"
    if dim == 1:
        # one dimension is where exact distance ties come from -- unit vectors are +-1, bf16 leaves ~250 distinct values
        # per binade -- and between equidistant nodes only recall is pinned, not ids
        dtype = "f32"
        if metric == "COSINE":
            metric = "L2"
    tag = (dim, n, M, efc, ef, k, nq, metric, dtype)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == "COSINE":
        x = np.stack([oracle.normalize(v)[0] for v in x])
    labels = rng.permutation(n).astype(np.uint64) + int(rng.integers(0, 500))
    g = vsa.Index("HNSW", dim, metric, initial_cap=n, m=M, ef_construction=efc, ef_runtime=10, build_threads=1, dtype=dtype)
    o = oracle.HNSW(dim, metric, max_elements=n, M=M, ef_construction=efc, ef=10)
    xo = _bf16_round(x) if dtype == "bf16" else x
    for i in range(n):
        assert g.add(int(labels[i]), x[i]) == 0
    o.add_many(xo, labels)
    st = g.stats()
    assert (st.count, st.max_level, st.entry_point) == (n, o.max_level, o.entry_point), tag
    if rng.random() < 0.3:
        for lab in rng.choice(labels, size=n // 8, replace=False):
            assert g.remove(int(lab)) == 0 and o.mark_delete(int(lab)) == 0
    allow = nbits = None
    if rng.random() < 0.3:
        nbits = int(labels.max()) + 1
        allow = oracle.allow_bitmap(labels[rng.random(n) < 0.3], nbits)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == "COSINE":
        Q = np.stack([oracle.normalize(v)[0] for v in Q])
    kw = {} if allow is None else {"allow": allow, "allow_nbits": nbits}
    D, L, N = g.search_batch(Q, k, ef=ef, **kw)
    for i in (range(nq) if nq <= 8 else sorted(rng.choice(nq, 8, replace=False).tolist())):
        od, ol = o.search(Q[i], k, ef=ef, **kw)
        assert N[i] == len(ol), tag
        assert L[i, :N[i]].tolist() == ol.tolist(), str(tag)
        assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), tag
    gd, gl = g.search(Q[0], k, ef=ef, **kw)           # the one-query entry point
    od, ol = o.search(Q[0], k, ef=ef, **kw)
    assert gl.tolist() == ol.tolist() and gd.view(np.uint32).tolist() == od.view(np.uint32).tolist(), tag
    if seed % 3 == 0:                                  # persistence round trip: the loaded index answers the same
        g2 = vsa.Index.load(g.save(), "HNSW", dim, metric, initial_cap=n, m=M, ef_construction=efc, dtype=dtype)
        D2, L2, N2 = g2.search_batch(Q, k, ef=ef, **kw)
        assert N2.tolist() == N.tolist() and L2.tolist() == L.tolist(), tag
        assert D2.view(np.uint32).tolist() == D.view(np.uint32).tolist(), tag
