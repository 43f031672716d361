"""Randomised parameter sweep over the FLAT path: index size, dimension, k, batch size, metric, row storage, filter
and deletions drawn from a fixed seed, so that every kernel-selection branch (scan with 1/2/4/8 queries per pass,
matrix-core path with 32/24/16-query tiles and register or HBM lists, selection merge and streaming merge, paging
for k > 1024) meets shapes nobody picked by hand.  Bar as everywhere: ids and distance bits equal to the CPU oracle."""
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


DIMS = [1, 3, 16, 17, 48, 64, 100, 128, 192, 255, 256, 320, 384, 512, 700, 768, 832, 960, 1024, 1100, 1536, 1600]
BATCHES = [1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 32, 33, 64, 100, 257]


@pytest.mark.parametrize("seed", range(72))
def test_random_shape(vsa, oracle, seed):
    rng = np.random.default_rng(5000 + seed + SWEEP_OFFSET)
    dim = int(rng.choice(DIMS))
    n = int(rng.integers(1, 60000 if dim <= 256 else 12000))
    nq = int(rng.choice(BATCHES))
    metric = str(rng.choice(["L2", "IP", "COSINE"]))
    dtype = "bf16" if rng.random() < 0.25 else "f32"
    k = int(min(n, rng.choice([1, 2, 5, 10, 10, 10, 33, 64, 65, 100, 256, 300, 1100])))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if rng.random() < 0.2:                       # duplicate rows: ties by label
        x[n // 2:] = x[: n - n // 2]
    if metric == "COSINE":
        x = np.stack([oracle.normalize(v)[0] for v in x])
    labels = (rng.permutation(n).astype(np.uint64) + int(rng.integers(0, 1000)))
    if rng.random() < 0.3 and n > 10:
        # grown the way the module grows it (vector_flat.cc:167-173): start small, resize by a block when full,
        # searches in between -- the row store is reallocated under the kernels several times
        cap = max(1, n // 5)
        g = vsa.Index("FLAT", dim, metric, initial_cap=cap, dtype=dtype)
        done = 0
        while done < n:
            room = cap - done
            if room == 0:
                cap = min(n, cap + max(1, n // 5))
                g.resize(cap)
                continue
            g.add_batch(x[done:done + room], labels[done:done + room])
            done += room
            g.search_batch(x[:min(8, done)], min(3, done))
    else:
        g = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype)
        g.add_batch(x, labels)
    o = oracle.Flat(dim, metric, max_elements=n)
    o.add_many(_bf16_round(x) if dtype == "bf16" else x, labels)
    removed = 0
    if n > 20 and rng.random() < 0.3:            # swap-deletes before the search
        for lab in rng.choice(labels, size=min(50, n // 4), replace=False):
            g.remove(int(lab))
            o.remove(int(lab))
            removed += 1
    k = min(k, n - removed)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    if nq > 1 and n > 3:
        Q[0] = x[3]                              # one query that is a stored row
    if metric == "COSINE":
        Q = np.stack([oracle.normalize(v)[0] for v in Q])
    allow = None
    nbits = None
    if rng.random() < 0.3:
        nbits = int(labels.max()) + 1
        keep = labels[rng.random(n) < rng.choice([0.5, 0.05])]
        allow = oracle.allow_bitmap(keep, nbits)
    if allow is None:
        D, L, N = g.search_batch(Q, k)
    else:
        D, L, N = g.search_batch(Q, k, allow=allow, allow_nbits=nbits)
    check = range(nq) if nq <= 16 else sorted(rng.choice(nq, 16, replace=False).tolist())
    for i in check:
        if allow is None:
            od, ol = o.search(Q[i], k)
            assert N[i] == len(ol), (dim, n, nq, metric, dtype, k)
            assert L[i, :N[i]].tolist() == ol.tolist(), (dim, n, nq, metric, dtype, k)
            assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), (dim, n, nq, metric, dtype, k)
        else:
            # with a filter the device returns the exact k best allowed rows by (distance, label); the oracle's
            # bruteforce can under-fill (bruteforce.h:128-141), so compare against the allowed rows ranked in full
            od, ol = o.search(Q[i], n - removed)
            bits = np.unpackbits(allow.view(np.uint8), bitorder="little")
            sel = [j for j, lab in enumerate(ol.tolist()) if lab < nbits and bits[lab]][:k]
            assert N[i] == len(sel), (dim, n, nq, metric, dtype, k)
            assert L[i, :N[i]].tolist() == ol[sel].tolist(), (dim, n, nq, metric, dtype, k)
            assert D[i, :N[i]].view(np.uint32).tolist() == od[sel].view(np.uint32).tolist()


@pytest.mark.parametrize("dim", [64, 576, 768])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097, 16383, 16385])
def test_sizes_around_tile_boundaries(vsa, oracle, n, dim):
    """Row counts one below, at and one above the tile sizes of the kernels (16 rows per wave in the scan, 128 per
    block on the matrix cores, 4096-entry merges), one query, a scan batch and a matrix-core batch; followed by a
    save / load round trip that must answer the same."""
    rng = np.random.default_rng(n * 131 + dim)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("FLAT", dim, "IP", initial_cap=n)
    g.add_batch(x)
    o = oracle.Flat(dim, "IP", max_elements=n)
    o.add_many(x)
    k = min(10, n)
    for nq in (1, 4, 40):
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        D, L, N = g.search_batch(Q, k)
        for i in range(0, nq, 7):
            od, ol = o.search(Q[i], k)
            assert N[i] == len(ol)
            assert L[i, :N[i]].tolist() == ol.tolist(), (n, dim, nq)
            assert D[i, :N[i]].view(np.uint32).tolist() == od.view(np.uint32).tolist(), (n, dim, nq)
    g2 = vsa.Index.load(g.save(), "FLAT", dim, "IP", initial_cap=n)
    D2, L2, N2 = g2.search_batch(Q, k)
    assert N2.tolist() == N.tolist() and L2.tolist() == L.tolist() and D2.view(np.uint32).tolist() == D.view(np.uint32).tolist()
