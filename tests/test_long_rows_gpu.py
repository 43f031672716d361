"""Row lengths up to the reference's default limit (max-vector-dimensions = 32768, ft_create_parser.cc:63): the LDS
query block of the scan, fewer waves per block in the HNSW search, the matrix-core path's tile choice (it hands
rows that long to the scan), and the device-assisted build over long rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("dim", [2500, 4096, 9000, 20000, 32768])
@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_flat_long_rows(vsa, oracle, dim, metric):
    rng = np.random.default_rng(dim)
    n = 300
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("FLAT", dim, metric, initial_cap=n)
    g.add_batch(x)
    o = oracle.Flat(dim, metric, max_elements=n)
    o.add_many(x)
    Q = rng.standard_normal((9, dim)).astype(np.float32)
    _same(*g.search(Q[0], 10), *o.search(Q[0], 10))
    D, L, N = g.search_batch(Q, 10)
    for i in range(len(Q)):
        _same(D[i], L[i], *o.search(Q[i], 10))


@pytest.mark.parametrize("dim", [2500, 9000, 12000, 32768])
def test_hnsw_long_rows(vsa, oracle, dim):
    rng = np.random.default_rng(dim + 1)
    n = 250
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=8, ef_construction=40, ef_runtime=32, build_threads=1)
    o = oracle.HNSW(dim, "L2", max_elements=n, M=8, ef_construction=40, ef=32)
    for i in range(n):
        assert g.add(i, x[i]) == 0
    o.add_many(x)
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    for ef in (32, 600):
        D, L, N = g.search_batch(Q, 10, ef=ef)
        for i in range(len(Q)):
            _same(D[i, :N[i]], L[i, :N[i]], *o.search(Q[i], 10, ef=ef))


@pytest.mark.parametrize("dim", [2048, 6000])
def test_device_build_long_rows(vsa, dim):
    rng = np.random.default_rng(dim + 2)
    n = 6000
    A = rng.standard_normal((dim, 16)).astype(np.float32)
    x = rng.standard_normal((n, 16)).astype(np.float32) @ A.T + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, ef_runtime=64)
    g.add_batch(x)
    f = vsa.Index("FLAT", dim, "L2", initial_cap=n)
    f.add_batch(x)
    Q = x[:100] + np.float32(0.01)
    D, L, N = g.search_batch(Q, 10, ef=64)
    Dt, Lt, Nt = f.search_batch(Q, 10)
    rec = np.mean([len(set(L[i, :N[i]].tolist()) & set(Lt[i].tolist())) / 10 for i in range(len(Q))])
    assert rec >= 0.9, rec
