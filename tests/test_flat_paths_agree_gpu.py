"""Mid-sized indexes of random shape: the matrix-core path (K4) and the scan (K3, VK_FLAT_FORCE_SCAN=1 at index
creation) must return the same ids and distance bits for the same batch -- the oracle is too slow at these sizes,
the two kernels share nothing but the arithmetic contract.  Sizes are drawn so that row-store allocations end in
many different places relative to the kernels' tiles (the overrun the sweep found depended on exactly that)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWEEP_OFFSET = int(__import__("os").environ.get("VK_SWEEP_OFFSET", "0"))


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


@pytest.mark.parametrize("seed", range(16))
def test_random_mid_size(vsa, monkeypatch, seed):
    rng = np.random.default_rng(4000 + seed + SWEEP_OFFSET)
    dim = int(rng.choice([64, 128, 200, 256, 576, 768, 1024, 1152]))
    n = int(rng.integers(20000, 400000 if dim <= 256 else 120000))
    nq = int(rng.choice([5, 8, 33, 64, 256]))
    k = int(rng.choice([1, 10, 10, 40, 100]))
    dtype = "bf16" if rng.random() < 0.25 else "f32"
    metric = str(rng.choice(["IP", "COSINE"]))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == "COSINE":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    monkeypatch.delenv("VK_FLAT_FORCE_SCAN", raising=False)
    a = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
    monkeypatch.setenv("VK_FLAT_FORCE_SCAN", "1")
    b = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
    monkeypatch.delenv("VK_FLAT_FORCE_SCAN", raising=False)
    a.add_batch(x)
    b.add_batch(x)
    allow = nbits = None
    if rng.random() < 0.3:
        nbits = n
        allow = np.zeros((n + 63) // 64, np.uint64)
        keep = np.flatnonzero(rng.random(n) < 0.2)
        np.bitwise_or.at(allow, keep >> 6, np.uint64(1) << (keep & 63).astype(np.uint64))
    kw = {} if allow is None else {"allow": allow, "allow_nbits": nbits}
    Da, La, Na = a.search_batch(Q, k, **kw)
    Db, Lb, Nb = b.search_batch(Q, k, **kw)
    tag = (dim, n, nq, k, dtype, metric, allow is not None)
    assert Na.tolist() == Nb.tolist(), tag
    assert La.tolist() == Lb.tolist(), tag
    assert Da.view(np.uint32).tolist() == Db.view(np.uint32).tolist(), tag
    d1, l1 = a.search(Q[0], k, **kw)                      # and the one-query path
    assert l1.tolist() == La[0, :Na[0]].tolist() and d1.view(np.uint32).tolist() == Da[0, :Na[0]].view(np.uint32).tolist(), tag
