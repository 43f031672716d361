import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import _pkg
from bench import gen_rows
from oracle import oracle as O
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
bits = O.allow_bitmap(np.flatnonzero(np.random.default_rng(1).random(N) < 0.1).astype(np.uint64), N)
for nq in (1, 16, 64, 256, 1024, 2048):
    h.search_batch(Q[:nq], 10, ef=128, allow=bits, allow_nbits=N)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): h.search_batch(Q[:nq], 10, ef=128, allow=bits, allow_nbits=N)
    dt = (time.perf_counter() - t0) / reps
    print(f"filtered nq={nq}: {dt*1e3:.3f} ms per batch, {nq/dt:.0f} QPS", flush=True)
