#!/usr/bin/env python3
"""Recall of the same data built three ways: device-assisted, host 16 threads, host 1 thread (= hnswlib's
sequential order)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
vsa = _pkg.vsa
n, dim, rank = int(sys.argv[1]), int(sys.argv[2]), 32
rng = np.random.default_rng(1234)
A = rng.standard_normal((dim, rank)).astype(np.float32)
def gen(m, seed):
    r = np.random.default_rng(seed)
    x = r.standard_normal((m, rank)).astype(np.float32) @ A.T + 0.05 * r.standard_normal((m, dim)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
X, Q = gen(n, 1), gen(2000, 2)
flat = vsa.Index("FLAT", dim, "IP", initial_cap=n)
flat.add_batch(X)
Dt, Lt, Nt = flat.search_batch(Q, 10)
runs = [("device", "1", 0), ("host-16", "0", 16)] if os.environ.get("CMP_QUICK") else [("device", "1", 0), ("host-16", "0", 16), ("host-1", "0", 1), ("device", "1", 0)]
for tag, dev, thr in runs:
    os.environ["VK_HNSW_DEVICE_BUILD"] = dev
    g = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128, build_threads=thr)
    t0 = time.time(); g.add_batch(X); g.flush(); dt = time.time() - t0
    out = f"{tag}: {dt:.1f}s"
    for ef in (32, 64, 128):
        D, L, N = g.search_batch(Q, 10, ef=ef)
        out += f" | ef={ef}: {np.mean([len(set(L[i, :N[i]].tolist()) & set(Lt[i].tolist())) / 10 for i in range(len(Q))]):.4f}"
    print(out, flush=True)
