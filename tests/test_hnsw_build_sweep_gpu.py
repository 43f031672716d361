"""Randomised sweep over the device-assisted HNSW build (K9): size, dimension, M, efConstruction, metric, row
storage and how the rows arrive (one add_batch, two, or a host-built prefix first).  Pinned: recall against the
exact answer no worse than the host build's, the structural rules LoadIndex checks (via a save/load round trip),
the degree bound, and that the CPU oracle walking the saved graph returns the device's answers."""
import numpy as np
import pytest

from test_hnsw_build_gpu import build, latent, recall

pytestmark = pytest.mark.gpu

# VK_SWEEP_OFFSET=<n> shifts every seed: a different set of shapes for a one-off hunt
SWEEP_OFFSET = int(__import__("os").environ.get("VK_SWEEP_OFFSET", "0"))


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


GAPS = []                  # (seed, shape, K9 recall, host recall) of the f32 shapes, filled by test_random_build
PER_SHAPE_SPREAD = 0.015


@pytest.mark.parametrize("seed", range(10))
def test_random_build(vsa, oracle, seed):
    rng = np.random.default_rng(7000 + seed + SWEEP_OFFSET)
    dim = int(rng.choice([16, 48, 100, 256, 768]))
    n = int(rng.integers(6000, 26000 if dim <= 100 else 12000))
    M = int(rng.choice([4, 8, 16, 32, 48]))
    efc = int(rng.choice([40, 100, 200]))
    metric = str(rng.choice(["IP", "L2"]))
    dtype = "bf16" if rng.random() < 0.3 else "f32"
    tag = (dim, n, M, efc, metric, dtype)
    x = latent(n, dim, 100 + seed)
    Q = latent(1024, dim, 200 + seed)
    flat = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
    flat.add_batch(x)
    how = int(rng.integers(0, 3))
    gd = vsa.Index("HNSW", dim, metric, initial_cap=n // 2 if how == 1 else n, m=M, ef_construction=efc, ef_runtime=64, dtype=dtype)
    if how == 0:
        gd.add_batch(x)
    elif how == 1:                                     # two bulk batches, the index resized in between (searched, too)
        cut = n // 2
        gd.add_batch(x[:cut], np.arange(cut, dtype=np.uint64))
        gd.search_batch(Q[:16], 10)
        gd.resize(n)
        gd.add_batch(x[cut:], np.arange(cut, n, dtype=np.uint64))
    else:                                              # a host-built prefix, then the bulk path
        for i in range(300):
            assert gd.add(i, x[i]) == 0
        gd.add_batch(x[300:], np.arange(300, n, dtype=np.uint64))
    assert gd.stats().count == n, tag
    gh = build(vsa, x, False, metric, M=M, efc=efc) if dtype == "f32" else None
    rd = recall(gd, flat, Q)
    if gh is not None:
        rh = recall(gh, flat, Q)
        GAPS.append((seed, tag, rd, rh))
        print("sweep seed %d %s: K9 %.4f host %.4f (%+.4f)" % (seed, tag, rd, rh, rd - rh))
        # one shape, 1024 queries: the spread between two builds of the same rows (different insertion interleavings) at
        # these sizes; the bar itself -- the mean over the sweep -- is test_sweep_mean_recall_matches_host below
        assert rd >= rh - PER_SHAPE_SPREAD, (tag, rd, rh)
    assert rd >= (0.55 if M == 4 else 0.8), (tag, rd)
    chunks = gd.save()
    g2 = vsa.Index.load(chunks, "HNSW", dim, metric, initial_cap=n, m=M, ef_construction=efc, dtype=dtype)
    assert g2.stats().count == n, tag
    M0 = 2 * M
    deg = np.array([int(np.frombuffer(c[:4], np.uint32)[0] & 0xFFFF) for c in chunks[1:1 + n]])
    assert deg.max() <= M0 and (deg == 0).sum() == 0, (tag, deg.max(), int((deg == 0).sum()))   # hnswlib never leaves a node without out-links
    if dtype == "f32":
        o = oracle.HNSW.from_saved_chunks(chunks, dim, metric, M, ef_construction=efc)
        for q in Q[:10]:
            d0, l0 = gd.search(q, 10, ef=64)
            d1, l1 = o.search(q, 10, ef=64)
            assert l0.tolist() == l1.tolist() and d0.view(np.uint32).tolist() == d1.view(np.uint32).tolist(), tag


def test_sweep_mean_recall_matches_host():
    """north_star: recall >= reference at identical ef -- over the sweep's f32 shapes the mean recall of the K9 graphs is
    within half a percent of the host-order graphs' (runs after test_random_build in file order)."""
    if len(GAPS) < 5:
        pytest.skip("runs behind test_random_build")
    rd, rh = np.mean([g[2] for g in GAPS]), np.mean([g[3] for g in GAPS])
    print("sweep: mean K9 %.4f, mean host %.4f over %d shapes" % (rd, rh, len(GAPS)))
    assert rd >= rh - 0.005, GAPS
