#!/usr/bin/env python3
"""BASELINE.json configs[0]: FLAT 100k x 128 f32 L2 k=10, one query per call (host in, host out through the
C ABI), device vs the CPU oracle on one thread, with parity on every query."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
from oracle import oracle as O
vsa = _pkg.vsa
n, dim, k, nq = 100_000, 128, 10, 1000
rng = np.random.default_rng(1234)
x = rng.standard_normal((n, dim)).astype(np.float32)
Q = np.random.default_rng(1235).standard_normal((nq, dim)).astype(np.float32)
g = vsa.Index("FLAT", dim, "L2", initial_cap=n)
g.add_batch(x)
o = O.Flat(dim, "L2", max_elements=n)
o.add_many(x)
g.search(Q[0], k)
t0 = time.perf_counter()
G = [g.search_one(q, k) for q in Q]
tg = (time.perf_counter() - t0) / nq
t0 = time.perf_counter()
C = [o.search(q, k) for q in Q[:200]]
tc = (time.perf_counter() - t0) / 200
same = all(a[1].tolist() == b[1].tolist() and a[0].view(np.uint32).tolist() == b[0].view(np.uint32).tolist() for a, b in zip(G, C))
D, L, N = g.search_batch(Q, k)
tb0 = time.perf_counter(); g.search_batch(Q, k); tb = time.perf_counter() - tb0
print(f"config 1 (FLAT {n}x{dim} L2 k={k}): device {tg*1e6:.0f} us/query ({1/tg:.0f} QPS, one caller), "
      f"CPU oracle ({O.cpu_path()}) {tc*1e3:.2f} ms/query on one thread; 200/200 identical: {same}; "
      f"batch of {nq}: {tb*1e3:.2f} ms = {nq/tb:.0f} QPS")
