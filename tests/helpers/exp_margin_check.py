"""Margin audit of the candidate filter -- run by tests/test_flat_filter_adversarial_gpu.py in its own process with
VKINDEX_LIB = libvkindex_exp.so (the -DVK_EXPERIMENTS build, where the final pass can dump what its gate saw:
vk_exp_filter_dump).  On the adversarial index of tests/helpers/adversarial.py, for the first DUMP rows and every query:

  1. the ERROR MODEL holds: |approx - exact| <= E_q(R_t) for every dumped (row, query) pair, exact = the f64 dot product of
     the f32 inputs (for L2: dot - |x|^2 / 2), E_q(R_t) recovered from the kernel's own threshold and bound;
  2. L_q is a lower bound of the k-th best EXACT score of every query;
  3. every row of the true answer inside the dump region passed its gate: approx >= thr;
  4. the answer is bit-identical to the exact kernels' (and the worst use of the margin is printed).
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch  # noqa: E402  (torch's HIP runtime first, see tests/conftest.py)

if torch.cuda.is_available():
    torch.cuda.init()
import _pkg  # noqa: E402
import adversarial  # noqa: E402

vsa = _pkg.vsa
assert "libvkindex_exp" in str(vsa.LIB_PATH), vsa.LIB_PATH
lib = C.CDLL(str(vsa.LIB_PATH))
lib.vk_exp_filter_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.vk_exp_filter_dump.restype = None
dev = torch.device("cuda", 0)
DUMP, K = 4096, 10
small = {"filter-prepass-rows": 1024, "filter-min-rows": 32768}


def audit(dim, dtype, metric, seed, options=None, nq=64, level0=300):
    bf16 = dtype == "bf16"
    X, Q, owner, is_a = adversarial.build(seed, dim, nq, K, bf16, dump_rows=DUMP, level0=level0)
    n = X.shape[0]
    opts = dict(small)
    opts.update(options or {})
    ix = vsa.Index("FLAT", dim, metric, initial_cap=n, dtype=dtype, options=opts)
    ix.add_batch(X)
    scores = torch.full((DUMP, nq), float("nan"), device=dev)
    thr = torch.full((DUMP // 128, nq), float("nan"), device=dev)
    qstate = torch.full((6, nq), float("nan"), device=dev)
    lib.vk_exp_filter_dump(scores.data_ptr(), thr.data_ptr(), qstate.data_ptr(), DUMP, nq)
    D, L, N = ix.search_batch(Q, K)
    torch.cuda.synchronize()
    lib.vk_exp_filter_dump(None, None, None, 0, 0)
    st = ix.stats()
    assert st.last_filter_candidates > 0 and st.last_filter_fallback == 0, (st.last_filter_candidates, st.last_filter_fallback)
    ix.set_option("flat-filter", 0)
    De, Le, Ne = ix.search_batch(Q, K)
    assert ix.stats().last_filter_candidates == 0
    assert (N == Ne).all() and (L == Le).all() and (D.view(np.uint32) == De.view(np.uint32)).all(), "filter path != exact path"

    S, T, QS = scores.cpu().numpy().astype(np.float64), thr.cpu().numpy().astype(np.float64), qstate.cpu().numpy().astype(np.float64)
    assert np.isfinite(S).all() and np.isfinite(T).all(), "the final pass did not dump its gate"
    Lq, closed = QS[4], QS[3]
    assert (closed == 0).all()
    Xs = adversarial.bf16_round(X) if bf16 else X             # what the index holds
    xd, qd = Xs[:DUMP].astype(np.float64), Q.astype(np.float64)
    exact = xd @ qd.T                                          # [DUMP][nq], accumulator space
    if metric == "L2":
        exact = exact - 0.5 * (xd * xd).sum(1, keepdims=True)
    # E_q(R_t) as the kernel applied it: thr = (L_q - E) - 2^-21 max(1, |L_q|)
    E = (Lq[None, :] - T) - 2.0 ** -21 * np.maximum(1.0, np.abs(Lq))[None, :]
    assert (E > 0).all()
    Erow = np.repeat(E, 128, axis=0)
    err = np.abs(S - exact)
    use = float((err / Erow).max())
    assert (err <= Erow).all(), ("error model violated", use)
    # 2. L_q <= the k-th best exact score (the exact answer's distances, turned back into accumulator space)
    full = Xs.astype(np.float64) @ qd.T
    if metric == "L2":
        full = full - 0.5 * (Xs.astype(np.float64) ** 2).sum(1, keepdims=True)
    kth = np.sort(full, axis=0)[-K]
    assert (Lq <= kth + 1e-12).all(), ("L_q above the k-th best exact score", float((Lq - kth).max()))
    # 3. the true answer's rows inside the dump region passed their gates; how much room they had, in margins
    room = []
    for q in range(nq):
        for lab in Le[q]:
            if lab < DUMP:
                a, t = S[int(lab), q], T[int(lab) // 128, q]
                assert a >= t, ("a true neighbour below its gate", q, int(lab), a, t)
                room.append((a - t) / E[int(lab) // 128, q])
    n_a_in_answer = int(sum(is_a[int(l)] for q in range(nq) for l in Le[q]))
    if metric != "L2":                                        # the construction put A rows into the answers
        assert len(room) >= nq and n_a_in_answer >= nq, (len(room), n_a_in_answer)
    print(f"{metric} {dtype} D={dim} n={n} nq={nq} {options or ''}: error model used to {use:.3f} of E; true neighbours' room above the gate: "
          f"min {min(room) if room else float('nan'):.2f} E over {len(room)} rows ({n_a_in_answer} A rows in the answers); tightness (kth - L_q) / E: "
          f"median {float(np.median((kth - Lq) / E.mean(0))):.2f}; survivors per query {st.last_filter_candidates / nq:.0f}")
    return use


worst = 0.0
for dim, dtype, metric, extra, nq, level0 in (
        (768, "f32", "IP", None, 64, 300), (128, "f32", "IP", None, 64, 300), (768, "bf16", "IP", None, 64, 300),
        (768, "bf16", "IP", {"filter-bf16-mfma": 0}, 64, 300), (768, "f32", "IP", {"filter-bdma": 0}, 64, 300),
        (256, "f32", "L2", None, 64, 300), (768, "bf16", "IP", {"filter-row-dma": 0}, 64, 300),
        # a population large enough for the SAMPLE to hold dozens of rows of the best level: L_q right under the k-th best
        (768, "f32", "IP", None, 8, 6000), (768, "bf16", "IP", None, 8, 6000), (256, "f32", "L2", None, 8, 6000)):
    worst = max(worst, audit(dim, dtype, metric, 20260929 + dim + nq, extra, nq, level0))
print(f"margin audit ok: worst use of the error margin {worst:.3f}")
