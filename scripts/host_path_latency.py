#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-buffer entry points (vk_index_search_batch / vk_index_search):
host queries in, host results out, at BASELINE config 2 size."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
from bench import gen_rows, device_view
vsa = _pkg.vsa
N, D, B, K = 10_000_000, 768, 256, 10
dev = torch.device("cuda", 0)
ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
p, stride = ix.device_rows(N)
t = device_view(p, (N, stride // 4), dev)
for lo, x in gen_rows(0, N, D, dev):
    t[lo:lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
Q = np.ascontiguousarray(t[:B, :D].cpu().numpy())
ix.search_batch(Q, K); ix.search(Q[0], K)
t0 = time.perf_counter()
for _ in range(10):
    ix.search_batch(Q, K)
tb = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for i in range(20):
    ix.search(Q[i], K)
t1 = (time.perf_counter() - t0) / 20
print(f"host-in/host-out: batch of {B}: {tb*1e3:.3f} ms ({B/tb:.0f} QPS); single query: {t1*1e3:.3f} ms")
