#!/usr/bin/env python3
"""HNSW time-to-ready: device-assisted bulk build (K9) vs the host build, same data, plus the recall of
both graphs.  Usage: build_probe.py --rows 200000 --dim 768"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
vsa = _pkg.vsa

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--rank", type=int, default=32)
ap.add_argument("--host-rows", type=int, default=0, help="also build this many rows on the host for comparison (0 = same as --rows)")
ap.add_argument("--skip-host", action="store_true")
ap.add_argument("--ef", type=int, nargs="+", default=[64, 128, 256])
a = ap.parse_args()

rng = np.random.default_rng(1234)
A = rng.standard_normal((a.dim, a.rank)).astype(np.float32)
def gen(n, seed):
    r = np.random.default_rng(seed)
    out = np.empty((n, a.dim), np.float32)
    for lo in range(0, n, 100000):
        m = min(100000, n - lo)
        x = r.standard_normal((m, a.rank)).astype(np.float32) @ A.T + 0.05 * r.standard_normal((m, a.dim)).astype(np.float32)
        out[lo:lo + m] = x / np.linalg.norm(x, axis=1, keepdims=True)
    return out
X = gen(a.rows, 1)
Q = gen(1000, 2)
flat = vsa.Index("FLAT", a.dim, "IP", initial_cap=a.rows)
flat.add_batch(X)
Dt, Lt, Nt = flat.search_batch(Q, 10)

def run(tag, device, n):
    os.environ["VK_HNSW_DEVICE_BUILD"] = "1" if device else "0"
    g = vsa.Index("HNSW", a.dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128)
    t0 = time.time()
    g.add_batch(X[:n])
    g.flush()
    dt = time.time() - t0
    line = f"{tag}: {n} rows x {a.dim} in {dt:.1f} s = {n / dt:.0f} inserts/s, max_level {g.stats().max_level}"
    if n == a.rows:
        for ef in a.ef:
            D, L, N = g.search_batch(Q, 10, ef=ef)
            rec = np.mean([len(set(L[i, :N[i]].tolist()) & set(Lt[i].tolist())) / 10 for i in range(len(Q))])
            line += f" | recall@10 ef={ef}: {rec:.4f}"
    if os.environ.get("BUILD_PROBE_DEGREES"):
        chunks = g.save()
        deg = np.array([int(np.frombuffer(c[:4], np.uint32)[0] & 0xFFFF) for c in chunks[1:1 + n]])
        line += f" | level-0 degree mean {deg.mean():.2f} p10 {np.percentile(deg, 10):.0f} p50 {np.percentile(deg, 50):.0f} max {deg.max()} full {(deg == 32).mean():.3f}"
    print(line, flush=True)

run("device-assisted build", True, a.rows)
if not a.skip_host:
    run("host build (%d threads)" % len(os.sched_getaffinity(0)), False, a.host_rows or a.rows)
