"""A device allocation that cannot be served: the call fails with the reference's InternalError (hnswlib throws
"Not enough memory" from its constructors and from resizeIndex, bruteforce.h:44-48, hnswalg.h:129-135,758-777; the wrappers
turn every exception into absl::InternalError, vector_flat.cc:68-72,165-176, vector_hnsw.cc:102-106,186-197), the index keeps
what it had, and the thread's NEXT calls are not charged with the failure (the HIP runtime keeps a per-thread last error that
a later launch check would otherwise pick up)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIM = 16000              # 64 000-byte rows: a few million of them exceed the 288 GB of the device while the host side stays small
TOO_MANY = 8_000_000     # 512 GB of rows


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(gd, gl, od, ol):
    assert gl.tolist() == ol.tolist()
    assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist()


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_create_beyond_the_device_fails_cleanly(vsa, algo):
    with pytest.raises(vsa.VkError) as e:
        vsa.Index(algo, DIM, "L2", initial_cap=TOO_MANY, m=8, ef_construction=32)
    assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
    # ... and the next index of the same thread works
    rng = np.random.default_rng(1)
    x = rng.standard_normal((500, 64)).astype(np.float32)
    g = vsa.Index(algo, 64, "L2", initial_cap=500, m=8, ef_construction=32)
    g.add_batch(x)
    d, l = g.search(x[3], 1)
    assert l.tolist() == [3] and d[0] == 0.0


@pytest.mark.parametrize("algo", ["FLAT", "HNSW"])
def test_resize_beyond_the_device_keeps_the_index(vsa, oracle, algo):
    rng = np.random.default_rng(2)
    n, k = 2000, 10
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    Q = rng.standard_normal((8, DIM)).astype(np.float32)
    g = vsa.Index(algo, DIM, "L2", initial_cap=n, m=8, ef_construction=32, ef_runtime=64)
    g.add_batch(x)
    g.flush()
    before = [g.search(q, k) for q in Q]
    bd, bl, bn = g.search_batch(Q, k)
    with pytest.raises(vsa.VkError) as e:
        g.resize(TOO_MANY)
    assert e.value.code == vsa.VK_ERR_INTERNAL, e.value
    st = g.stats()
    assert st.count == n and st.capacity < TOO_MANY
    # the very next calls of this thread: single query, batch (the matrix-core path for FLAT), a write, a search after it
    for q, (d0, l0) in zip(Q, before):
        _same(*g.search(q, k), d0, l0)
    d, l, c = g.search_batch(Q, k)
    assert c.tolist() == bn.tolist() and l.tolist() == bl.tolist() and d.view(np.uint32).tolist() == bd.view(np.uint32).tolist()
    assert g.add(n, x[0]) == vsa.VK_ERR_CAPACITY       # (full: the caller resizes -- by a size that fits -- and retries)
    g.resize(n + 16)
    assert g.add(n, x[0]) == vsa.VK_OK
    d, l = g.search(x[0], 2)
    assert sorted(l.tolist()) == [0, n] and d.tolist() == [0.0, 0.0]
    if algo == "FLAT":
        o = oracle.Flat(DIM, "L2", max_elements=n + 16)
        o.add_many(np.concatenate([x, x[:1]]))
        for q in Q[:3]:
            _same(*g.search(q, k), *o.search(q, k))
