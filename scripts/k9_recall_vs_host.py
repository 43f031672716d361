#!/usr/bin/env python3
"""K9 (device-assisted bulk build, hnsw_build.hip) against the host builder on the same rows: recall@10 against the exact
answer at several ef, over several data seeds and >= 2048 queries each; level-0 out-degree statistics of both graphs (a node
without out-links is something hnswlib never leaves behind).

  python scripts/k9_recall_vs_host.py ROWS DIM [SEEDS=3] [QUERIES=2048] [EF,EF,...] [host|nohost]

Data: BASELINE.json's rank-32 latent model, L2-normalised (SURVEY 8(d) Config 2), M = 16, efConstruction = 200, COSINE."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg  # noqa: E402

vsa = _pkg.vsa
n, dim = int(sys.argv[1]), int(sys.argv[2])
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
efs = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [64, 128, 256]
with_host = (sys.argv[6] if len(sys.argv) > 6 else "host") == "host"
rank, M, efc = 32, 16, 200


def gen(A, m, seed, chunk=1 << 18):
    out = np.empty((m, dim), np.float32)
    r = np.random.default_rng(seed)
    for lo in range(0, m, chunk):
        hi = min(m, lo + chunk)
        x = r.standard_normal((hi - lo, rank)).astype(np.float32) @ A.T + 0.05 * r.standard_normal((hi - lo, dim)).astype(np.float32)
        out[lo:hi] = x / np.linalg.norm(x, axis=1, keepdims=True)
    return out


def degrees(g, count):
    """level-0 out-degree of every element, read off the index's own SaveIndex stream (the first word of an element chunk)"""
    esz = (2 * M + 1) * 4 + dim * 4 + 8
    deg = np.zeros(count, np.uint16)
    state = {"i": 0, "hdr": True}

    @vsa.WRITE_CHUNK
    def wr(_u, data, nbytes):
        if state["hdr"]:
            state["hdr"] = False
            return 0
        if state["i"] < count and nbytes == esz:
            deg[state["i"]] = C.cast(data, C.POINTER(C.c_uint32))[0] & 0xFFFF
            state["i"] += 1
        return 0

    rc = vsa.lib().vk_index_save(g._h, wr, None)
    assert rc == 0 and state["i"] == count, (rc, state["i"], count)
    return deg


def build(X, device, threads=0):
    os.environ["VK_HNSW_DEVICE_BUILD"] = "1" if device else "0"
    g = vsa.Index("HNSW", dim, "COSINE", initial_cap=len(X), m=M, ef_construction=efc, ef_runtime=128, build_threads=threads)
    t0 = time.time()
    g.add_batch(X)
    g.flush()
    return g, time.time() - t0


def recalls(g, Q, truth):
    out = {}
    for ef in efs:
        hit = 0
        for lo in range(0, len(Q), 1024):
            D, L, N = g.search_batch(Q[lo:lo + 1024], 10, ef=ef)
            hit += sum(len(set(L[i, :N[i]].tolist()) & set(truth[lo + i].tolist())) for i in range(L.shape[0]))
        out[ef] = hit / (10.0 * len(Q))
    return out


rows = []
for sd in range(seeds):
    A = np.random.default_rng(1234 + sd).standard_normal((dim, rank)).astype(np.float32)
    X, Q = gen(A, n, 100 + sd), gen(A, nq, 900 + sd)
    flat = vsa.Index("FLAT", dim, "COSINE", initial_cap=n)
    flat.add_batch(X)
    truth = np.concatenate([flat.search_batch(Q[lo:lo + 256], 10)[1] for lo in range(0, nq, 256)])
    del flat
    rec = {"seed": sd, "rows": n, "dim": dim, "queries": nq}
    for tag, device in (("k9", True),) + ((("host", False),) if with_host else ()):
        g, dt = build(X, device)
        st = g.stats()
        deg = degrees(g, st.count)
        rec[tag] = {"build_s": round(dt, 1), "recall": {str(k): round(v, 5) for k, v in recalls(g, Q, truth).items()},
                    "deg0": int((deg == 0).sum()), "deg_min": int(deg.min()), "deg_mean": round(float(deg.mean()), 2),
                    "deg_max": int(deg.max()), "staged_on_device": int(st.staged_adds_device)}
        del g
    print(json.dumps(rec), flush=True)
    rows.append(rec)
summary = {"rows": n, "dim": dim, "seeds": seeds, "queries_per_seed": nq}
for tag in ("k9", "host"):
    if tag in rows[0]:
        summary[tag] = {"mean_recall": {str(ef): round(float(np.mean([r[tag]["recall"][str(ef)] for r in rows])), 5) for ef in efs},
                        "deg0_total": sum(r[tag]["deg0"] for r in rows), "build_s": [r[tag]["build_s"] for r in rows]}
if "host" in summary:
    summary["k9_minus_host"] = {str(ef): round(summary["k9"]["mean_recall"][str(ef)] - summary["host"]["mean_recall"][str(ef)], 5) for ef in efs}
print(json.dumps({"summary": summary}), flush=True)
