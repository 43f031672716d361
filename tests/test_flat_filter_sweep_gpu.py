"""Randomised sweep over the f16 candidate filter + exact re-rank (K4h): index size, dimension (padding, 1 to 25 pipeline
stages per tile), batch size around the 32-query tiles and the 256-query launch groups, k up to 70, metric, row storage,
row scales from 1e-7 to 1e2 (f16 subnormals and zeros included), mixed-scale rows, duplicated rows (ties by label, survivor
lists continued in spill chunks), allow-bitmaps and deletions, drawn from a fixed seed.  The filter path must return exactly what the oracle returns -- ids and distance bits -- whatever it did on
the way (vk_index_stats says whether it re-ranked its survivors or handed the batch to the exact kernel)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SWEEP_OFFSET = int(os.environ.get("VK_SWEEP_OFFSET", "0"))


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _bf16_round(x):
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


DIMS = [8, 33, 64, 65, 100, 128, 200, 256, 384, 500, 768, 1024, 1536]
BATCHES = [33, 40, 63, 64, 65, 96, 128, 200, 255, 256, 257, 300, 513]


@pytest.mark.parametrize("seed", range(64))
def test_random_shape_through_the_filter(vsa, oracle, seed):
    rng = np.random.default_rng(9000 + seed + SWEEP_OFFSET)
    dim = int(rng.choice(DIMS))
    n = int(rng.integers(34_000, 70_000 if dim <= 512 else 40_000))
    # every fourth seed (small dimensions): three times the rows and the two-pass pipeline switched on for indexes of two tiles
    # per CU -- an early pass over up to half of every block's range, the main pass's bound from its survivors (r05)
    two_pass = seed % 4 == 3 and dim <= 256
    if two_pass:
        n *= 3
    nq = int(rng.choice(BATCHES))
    metric = str(rng.choice(["L2", "IP", "COSINE"]))
    dtype = "bf16" if rng.random() < 0.3 else "f32"
    k = int(rng.choice([1, 3, 10, 10, 10, 17, 32, 64, 70]))
    scale = float(10.0 ** rng.uniform(-3, 2))
    # seeds from 40 on: the f16 pipe's small end.  Row scales down to 1e-7 (elements that are f16 subnormals -- below
    # 6.1e-5 -- or round to zero), rows of mixed scale inside one tile, queries with elements below 6.1e-5: the margin's
    # absolute term (2^-25 per element) must cover what the conversion and the matrix core make of them.
    tiny = seed >= 40
    if tiny:
        scale = float(10.0 ** rng.uniform(-7, -4))
    nc = int(rng.integers(5, 60))
    centres = rng.standard_normal((nc, dim)).astype(np.float32)
    spread = float(rng.uniform(0.05, 1.0))
    x = (centres[rng.integers(0, nc, n)] + spread * rng.standard_normal((n, dim)).astype(np.float32)) * np.float32(scale)
    if tiny and seed % 3 == 0:                    # mixed scales: every row its own factor between 1 and 1e6
        x = (x * (10.0 ** rng.uniform(0, 6, (n, 1))).astype(np.float32)).astype(np.float32)
    if tiny and seed % 4 == 1:                    # some elements exactly in the subnormal band next to ordinary ones
        x[:, : dim // 2] = (x[:, : dim // 2] / np.float32(scale) * np.float32(3e-6)).astype(np.float32)
    dup = rng.random()
    if dup < 0.15:                                # a few hundred copies: ties at the k-th distance
        x[1000:1400] = x[999]
    elif dup < 0.25:                              # tens of thousands: the survivor lists overflow
        x[n // 3:] = x[7]
    if metric == "COSINE":
        x = (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    labels = rng.permutation(3 * n)[:n].astype(np.uint64)
    old = {v: os.environ.get(v) for v in ("VK_FILTER_PREPASS", "VK_FILTER_MIN_ROWS", "VK_FILTER_SEED")}
    # odd seeds: the sample's own k-th best comes from a filter pass seeded by the exact kernel over its first 128 rows
    # per 10 of k (what a large index does); even seeds: from the exact kernel over the whole sample
    os.environ.update(VK_FILTER_PREPASS="1024", VK_FILTER_MIN_ROWS="32768", VK_FILTER_SEED="128" if seed % 2 else "1000000")
    try:
        g = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
    finally:
        for v, o in old.items():
            if o is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = o
    if two_pass:
        g.set_option("filter-two-pass-min-tiles", 2)
    g.add_batch(x, labels)
    xs = _bf16_round(x) if dtype == "bf16" else x
    o = oracle.Flat(dim, metric, max_elements=n)
    o.add_many(xs, labels)
    if rng.random() < 0.3:
        for lab in rng.choice(labels, 500, replace=False):
            g.remove(int(lab))
            o.remove(int(lab))
    qscale = scale if not tiny or seed % 2 == 0 else float(10.0 ** rng.uniform(-6, 0))
    Q = (centres[rng.integers(0, nc, nq)] + spread * rng.standard_normal((nq, dim)).astype(np.float32)) * np.float32(qscale)
    if metric == "COSINE":
        Q = (Q / np.maximum(np.linalg.norm(Q, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    allow = nbits = None
    if rng.random() < 0.3:
        nbits = int(labels.max()) + 1
        allow = oracle.allow_bitmap(labels[rng.random(n) < rng.choice([0.5, 0.1, 0.01])], nbits)
    D, L, N = g.search_batch(Q, k, allow=allow, allow_nbits=nbits)
    st = g.stats()
    if two_pass and st.last_filter_candidates > 0 and st.count >= 256 * 2 * 128:
        assert 0 < st.last_filter_final_rows < st.count, (st.last_filter_final_rows, st.count)   # the batch did take two passes
    # (the path needs an index at least eight times the bound's sample: 1024 rows per 10 of k with this test's settings)
    # ... and rows the f16 pipe can carry: an index that is mostly tiles with a value beyond 32768 (or, for L2, a half
    # norm beyond f16) is kept off the path altogether
    f16_rows = float(np.abs(xs).max()) <= 32768.0 and (metric != "L2" or float((xs.astype(np.float64) ** 2).sum(1).max()) <= 1.0e5)
    if g.stats().count >= 8 * 1024 * ((k + 9) // 10) and f16_rows:
        assert st.last_filter_candidates > 0 or st.last_filter_fallback >= 1, "the batch did not take the filter path: %s %s dim %d n %d nq %d k %d scale %g count %d allow %s" % (
            metric, dtype, dim, n, nq, k, scale, g.stats().count, allow is not None)
    if allow is not None:
        # with a filter the reference's brute-force loop can under-fill (bruteforce.h:120-141, unreachable through
        # FT.SEARCH); the product returns the exact k best allowed rows: the oracle over the allowed rows alone
        keep = np.array([bool((allow[int(l) >> 6] >> np.uint64(int(l) & 63)) & np.uint64(1)) for l in labels])
        live = np.array([o.distance(int(l), Q[0]) is not None for l in labels])
        o = oracle.Flat(dim, metric, max_elements=n)
        o.add_many(xs[keep & live], labels[keep & live])
    for i in rng.choice(nq, min(nq, 24), replace=False):
        od, ol = o.search(Q[i], k)
        assert N[i] == len(ol), (seed, i)
        assert L[i, :N[i]].tolist() == ol.tolist(), (seed, i, metric, dtype, dim, n, nq, k)
        assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), (seed, i)
