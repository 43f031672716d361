"""bf16 row storage (extension; the reference only has FLOAT32, so parity is against the oracle's f32
path over the SAME rounded rows): rows arrive as f32, are rounded to nearest-even bf16 at ingest, and the
distance is the lane-exact f32 arithmetic on the widened values -- ids and distance bits must equal an
f32 oracle index holding the rounded rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("metric", ["L2", "IP"])
@pytest.mark.parametrize("n,dim", [(8000, 128), (3000, 100), (2500, 768)])
def test_flat_bf16_equals_f32_on_rounded_rows(vsa, oracle, metric, n, dim):
    rng = np.random.default_rng(41)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype="bf16")
    g.add_batch(x)
    assert g.stats().device_bytes < n * dim * 4          # half-width rows
    xr = bf16_round(x)
    o = oracle.Flat(dim, metric, max_elements=n)
    o.add_many(xr)
    Q = rng.standard_normal((20, dim)).astype(np.float32)
    for q in Q[:4]:
        _same(*g.search(q, 10), *o.search(q, 10))
    D, L, N = g.search_batch(Q, 10)
    for i in range(len(Q)):
        _same(D[i, :N[i]], L[i, :N[i]], *o.search(Q[i], 10))
    assert np.array_equal(g.get_row(7), xr[7])
    labels = rng.permutation(n)[:300].astype(np.uint64)
    gd, gl = g.search_labels(Q[0], 5, labels)
    od, ol = oracle.prefilter_topk(metric, Q[0], xr[labels.astype(int)], labels, 5)
    _same(gd, gl, od, ol)


def test_hnsw_bf16_equals_f32_on_rounded_rows(vsa, oracle):
    rng = np.random.default_rng(42)
    n, dim = 3000, 64
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=16, ef_construction=100, dtype="bf16")
    for i in range(n):
        assert g.add(i, x[i]) == 0
    o = oracle.HNSW(dim, "L2", max_elements=n, M=16, ef_construction=100)
    o.add_many(bf16_round(x))
    Q = rng.standard_normal((8, dim)).astype(np.float32)
    for q in Q:
        _same(*g.search(q, 10, ef=64), *o.search(q, 10, ef=64))


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 64, 10), (6000, 768, 32, 10), (3000, 100, 40, 30)])
def test_flat_bf16_batched_mfma_path(vsa, oracle, n, dim, nq, k):
    """>= 5 queries in the inner-product space over bf16 rows: K4 widens the rows on their way into LDS;
    the answer must equal the f32 oracle over the rounded rows, ids and distance bits."""
    rng = np.random.default_rng(43)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("FLAT", dim, "IP", initial_cap=n, dtype="bf16")
    g.add_batch(x)
    o = oracle.Flat(dim, "IP", max_elements=n)
    o.add_many(bf16_round(x))
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    D, L, N = g.search_batch(Q, k)
    for i in range(nq):
        od, ol = o.search(Q[i], k)
        _same(D[i, :N[i]], L[i, :N[i]], od, ol)


def test_flat_bf16_save_load_round_trip(vsa):
    """A bf16 FLAT index written out and read back (as f32 rows, the chunk format of the reference) answers the same."""
    rng = np.random.default_rng(61)
    n, dim = 3000, 200
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("FLAT", dim, "L2", initial_cap=n, dtype="bf16")
    g.add_batch(x)
    Q = rng.standard_normal((9, dim)).astype(np.float32)
    D, L, N = g.search_batch(Q, 10)
    g2 = vsa.Index.load(g.save(), "FLAT", dim, "L2", initial_cap=n, dtype="bf16")
    D2, L2, N2 = g2.search_batch(Q, 10)
    assert L2.tolist() == L.tolist() and D2.view(np.uint32).tolist() == D.view(np.uint32).tolist() and N2.tolist() == N.tolist()
