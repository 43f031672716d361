"""Probe: HNSW build speed (host threads) and device search throughput/recall at a given size."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--ef", type=int, default=128)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--dtype", default="f32")
args = ap.parse_args()
dev = torch.device("cuda", 0)
N, D = args.rows, args.dim
x = torch.empty(N, D, device=dev)
for lo, c in gen_rows(0, N, D, dev):
    x[lo:lo + c.shape[0]] = c
hx = x.cpu().numpy()
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(args.nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(args.nq, D, generator=g, device=dev), dim=1).cpu().numpy()
t = time.time()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=args.ef, build_threads=args.threads, dtype=args.dtype)
h.add_batch(hx)
h.flush()
tb = time.time() - t
print(f"build {N}x{D}: {tb:.1f}s = {N/tb:.0f} inserts/s, threads={args.threads or os.cpu_count()}, maxlevel={h.stats().max_level}", flush=True)
f = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
f.add_batch(hx)
_, Lf, _ = f.search_batch(Q, 10)
for ef in (args.ef, 256):
    h.search_batch(Q[:64], 10, ef=ef)
    t = time.time()
    reps = 5
    for _ in range(reps):
        Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)
    dt = (time.time() - t) / reps
    st = h.stats()
    rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Lh, Lf)) / (10.0 * len(Q))
    useful = (st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132)
    print(f"ef={ef}: {len(Q)/dt:.0f} QPS ({dt*1e3:.2f} ms per {len(Q)}-batch), recall@10={rec:.4f}, "
          f"n_eval/q={st.last_n_eval/len(Q):.0f} hops/q={st.last_n_hops/len(Q):.0f}, useful {useful/dt/1e9:.0f} GB/s", flush=True)
t = time.time()
for i in range(50):
    h.search(Q[i], 10, ef=args.ef)
print(f"single-query latency {1e3*(time.time()-t)/50:.3f} ms")
