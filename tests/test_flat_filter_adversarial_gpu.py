"""The candidate filter (K4h) on inputs built to break it: "cannot drop a true neighbour" tested, not argued.

  * tests/helpers/adversarial.py: rows and queries at f16 / bf16 ROUNDING MIDPOINTS, the true neighbours rounded so that
    every product loses and the sample's witnesses so that every product gains, hundreds of rows within a fraction of one
    margin of the k-th best score -- through the product library, against the exact kernels and the oracle;
  * the margin audit (tests/helpers/exp_margin_check.py, experiments build): the kernel's own approximate scores,
    thresholds and bounds against exact arithmetic -- |approx - exact| <= E_q(R_t), L_q <= k-th best, true neighbours above
    their gates; prints how much of the margin the worst case used;
  * a sweep of 1 000+ seeds over small adversarial indexes (dimension, element scale, metric, row storage, k, batch size,
    near-equal norms for L2), filter path against the exact path bit for bit.
The matrix-core instruction's own accumulation error is measured in tests/test_mfma_error_gpu.py."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "helpers"))
SMALL = {"filter-prepass-rows": 1024, "filter-min-rows": 32768}


@pytest.fixture(scope="module")
def vsa():
    import _pkg
    return _pkg.vsa


def _same(a, b, what):
    (ad, al, an), (bd, bl, bn) = a, b
    assert an.tolist() == bn.tolist(), what
    assert (al == bl).all(), what
    assert (ad.view(np.uint32) == bd.view(np.uint32)).all(), what


@pytest.mark.parametrize("dim,dtype,metric", [(768, "f32", "IP"), (128, "f32", "COSINE"), (768, "bf16", "IP"), (256, "f32", "L2"),
                                              (320, "bf16", "L2")])
def test_rows_at_rounding_midpoints_aligned_with_the_query(vsa, oracle, dim, dtype, metric):
    import adversarial
    k, nq = 10, 64
    X, Q, owner, is_a = adversarial.build(4100 + dim, dim, nq, k, dtype == "bf16")
    if metric == "COSINE":       # the ABI's COSINE is the caller-normalised inner product: normalising moves the elements off
        X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)     # the midpoints -- a second, unaligned population
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    n = X.shape[0]
    ix = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype, options=SMALL)
    ix.add_batch(X)
    got = ix.search_batch(Q, k)
    st = ix.stats()
    assert st.last_filter_candidates > 0 and st.last_filter_fallback == 0
    ix.set_option("flat-filter", 0)
    exact = ix.search_batch(Q, k)
    assert ix.stats().last_filter_candidates == 0
    _same(got, exact, (dim, dtype, metric))
    if metric != "L2":   # (L2 ranks by |x|^2 - 2 x.q: the rows' norms, not their rounding, order the few best)
        # the true answers are made of the rows built to lose (A) -- the construction did what it is for
        assert sum(bool(is_a[int(l)]) for q in range(nq) for l in exact[1][q]) >= nq // 4
    xs = adversarial.bf16_round(X) if dtype == "bf16" else X
    o = oracle.Flat(dim, metric, max_elements=n)
    o.add_many(xs, np.arange(n, dtype=np.uint64))
    for i in range(0, nq, 4):
        od, ol = o.search(Q[i], k)
        assert got[1][i].tolist() == ol.tolist() and got[0][i].view(np.uint32).tolist() == od.view(np.uint32).tolist(), i
    # larger k and a batch that is not a multiple of the query tile, same index
    ix.set_option("flat-filter", 1)
    g2 = ix.search_batch(Q[:37], 40)
    assert ix.stats().last_filter_candidates > 0
    ix.set_option("flat-filter", 0)
    _same(g2, ix.search_batch(Q[:37], 40), "k=40")


def test_margin_audit_in_the_experiments_build():
    """Approximate scores, thresholds and bounds dumped by the final pass (experiments build only) against exact arithmetic."""
    import _pkg
    exp = _pkg.vsa.EXP_LIB_PATH
    if not exp.exists():
        _pkg.vsa.build_experiments()
    env = dict(os.environ, VKINDEX_LIB=str(exp))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "helpers" / "exp_margin_check.py")], cwd=ROOT, capture_output=True, text=True,
                       env=env, timeout=900)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "margin audit ok" in r.stdout


def _small_case(rng):
    """One small adversarial index: a handful of queries, their rows at rounding midpoints, near-equal norms."""
    import adversarial
    dim = int(rng.choice([64, 128, 192, 256, 320]))
    dtype = "bf16" if rng.random() < 0.3 else "f32"
    metric = str(rng.choice(["IP", "IP", "L2", "COSINE"]))
    k = int(rng.choice([1, 5, 10, 10, 32]))
    nq = int(rng.choice([8, 16, 33]))
    # element magnitudes from 2^-12 (squares near the f16 subnormals) to 2^6 (L2 half norms in the thousands)
    base_exp = int(rng.integers(-12, 7)) if metric != "COSINE" else None
    if metric == "L2" and base_exp is not None:
        base_exp = min(base_exp, 2)                               # (half norms beyond f16 take the tile off the path: other tests)
    per = max(1, 33000 // (nq * 20))
    X, Q, owner, is_a = adversarial.build(int(rng.integers(1 << 40)), dim, nq, k, dtype == "bf16", a_per_query=max(4, k // 2 + 2), level0=3 * per, levels=17,
                                          per_level=per, base_exp=base_exp, dump_rows=nq * max(4, k // 2 + 2), exact_flips=False)
    if metric == "COSINE":
        X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    return dim, dtype, metric, k, X, Q


@pytest.mark.parametrize("chunk", range(16))
def test_thousand_seeds_of_small_adversarial_indexes(vsa, chunk):
    """64 seeds per chunk, 1 024 in all."""
    took = 0
    for seed in range(chunk * 64, chunk * 64 + 64):
        rng = np.random.default_rng(770000 + seed)
        dim, dtype, metric, k, X, Q = _small_case(rng)
        n = X.shape[0]
        ix = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype, options=SMALL)
        ix.add_batch(X)
        got = ix.search_batch(Q, k)
        took += ix.stats().last_filter_candidates > 0
        assert ix.stats().last_filter_fallback == 0, seed
        ix.set_option("flat-filter", 0)
        _same(got, ix.search_batch(Q, k), (seed, dim, dtype, metric, k, n))
        del ix
    assert took >= 60, took                                       # (nearly) every case went through the filter
