import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import _pkg
from bench import gen_rows
vsa = _pkg.vsa
dev = torch.device("cuda", 0)
N, D = 1_000_000, 768
x = torch.empty(N, D, device=dev)
for lo, c in gen_rows(0, N, D, dev):
    x[lo:lo + c.shape[0]] = c
hx = x.cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=128)
h.add_batch(hx); h.flush()
Q = hx[:4096] + 0.01
for nq in (1, 16, 64, 256, 512, 1024, 2048, 4096):
    h.search_batch(Q[:nq], 10, ef=128)
    t0 = time.perf_counter(); reps = 10 if nq < 1024 else 3
    for _ in range(reps): h.search_batch(Q[:nq], 10, ef=128)
    dt = (time.perf_counter() - t0) / reps
    print(f"nq={nq}: {dt*1e3:.3f} ms per batch, {nq/dt:.0f} QPS", flush=True)
