"""Randomised sweep over the pre-filter path (vk_index_search_labels = VectorBase::AddPrefilteredKey,
vector_base.cc:509-530): key lists of every length, with unknown labels mixed in, over FLAT and HNSW indexes of
random shape, after deletions.  The answer is the reference's heap rule over the keys in the order given: fill to
k, then replace the top only on a strictly smaller distance (ties keep the earlier key)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# VK_SWEEP_OFFSET=<n> shifts every seed: a different set of shapes for a one-off hunt
SWEEP_OFFSET = int(__import__("os").environ.get("VK_SWEEP_OFFSET", "0"))


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


@pytest.mark.parametrize("seed", range(24))
def test_random_key_lists(vsa, oracle, seed):
    rng = np.random.default_rng(3000 + seed + SWEEP_OFFSET)
    algo = "HNSW" if seed % 3 == 0 else "FLAT"
    dim = int(rng.choice([1, 16, 100, 128, 384, 768, 1100]))
    n = int(rng.integers(20, 3000 if algo == "FLAT" else 600))
    metric = str(rng.choice(["L2", "IP", "COSINE"]))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if rng.random() < 0.3:
        x[n // 2:] = x[: n - n // 2]                       # duplicate rows: ties
    if metric == "COSINE":
        x = np.stack([oracle.normalize(v)[0] for v in x])
    labels = rng.permutation(n).astype(np.uint64) + 10
    kw = dict(m=8, ef_construction=40) if algo == "HNSW" else {}
    g = vsa.Index(algo, dim, metric, initial_cap=n, **kw)
    g.add_batch(x, labels)
    live = {int(l): i for i, l in enumerate(labels)}
    if rng.random() < 0.4:
        for lab in rng.choice(labels, size=n // 5, replace=False):
            g.remove(int(lab))
            del live[int(lab)]
    for _ in range(6):
        m = int(rng.choice([0, 1, 2, 9, 10, 11, 100, 1000, n]))
        pool = np.concatenate([labels, np.arange(10 ** 6, 10 ** 6 + 50, dtype=np.uint64)])   # some unknown labels
        keys = rng.choice(pool, size=min(m, len(pool)), replace=False).astype(np.uint64)
        k = int(rng.choice([1, 5, 10, 64]))
        q = rng.standard_normal(dim).astype(np.float32)
        if metric == "COSINE":
            q = oracle.normalize(q)[0]
        gd, gl = g.search_labels(q, k, keys)
        known = [int(l) for l in keys.tolist() if int(l) in live]
        rows = x[[live[l] for l in known]] if known else np.zeros((0, dim), np.float32)
        od, ol = oracle.prefilter_topk(metric, q, rows, np.array(known, dtype=np.uint64), k)
        assert gl.tolist() == ol.tolist(), (algo, dim, n, metric, m, k)
        assert gd.view(np.uint32).tolist() == od.view(np.uint32).tolist(), (algo, dim, n, metric, m, k)
