#!/usr/bin/env python3
"""FLAT L2 (the metric that cannot use the matrix cores) at BASELINE config-2 size: batch throughput of K3."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
from bench import gen_rows, device_view
vsa = _pkg.vsa
N, D, K = 10_000_000, 768, 10
dev = torch.device("cuda", 0)
ix = vsa.Index("FLAT", D, "L2", initial_cap=N)
p, stride = ix.device_rows(N)
t = device_view(p, (N, stride // 4), dev)
for lo, x in gen_rows(0, N, D, dev):
    t[lo:lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
Q = np.ascontiguousarray(t[:256, :D].cpu().numpy()) + np.float32(0.01)
for B in (8, 64, 256):
    ix.search_batch(Q[:B], K)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): ix.search_batch(Q[:B], K)
    dt = (time.perf_counter() - t0) / reps
    print(f"L2 B={B}: {dt*1e3:.2f} ms, {B/dt:.0f} QPS, {3*N*D*B/dt/1e12:.1f} TFLOP/s (sub+mul+add), {N*D*B/dt/1e12:.1f} T element-pairs/s", flush=True)
