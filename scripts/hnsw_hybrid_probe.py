import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkg
from bench import gen_rows
vsa = _pkg.vsa
dev = torch.device("cuda", 0)
N, D = 2_000_000, 768
x = torch.empty(N, D, device=dev)
for lo, c in gen_rows(0, N, D, dev):
    x[lo:lo + c.shape[0]] = c
hx = x.cpu().numpy()
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(1024, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(1024, D, generator=g, device=dev), dim=1).cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=256)
t = time.time(); h.add_batch(hx); h.flush(); print(f"build {time.time()-t:.1f}s", flush=True)
f = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
f.add_batch(hx)
for sel in (0.1, 0.01):
    keep = np.flatnonzero(np.random.default_rng(5).random(N) < sel)
    bits = np.zeros((N + 63) // 64, np.uint64)
    np.bitwise_or.at(bits, keep >> 6, np.uint64(1) << (keep & 63).astype(np.uint64))
    _, Lf, _ = f.search_batch(Q, 10, allow=bits, allow_nbits=N)
    h.search_batch(Q[:64], 10, ef=256, allow=bits, allow_nbits=N)
    t = time.time(); Dh, Lh, Nh = h.search_batch(Q, 10, ef=256, allow=bits, allow_nbits=N); dt = time.time() - t
    st = h.stats()
    rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(Lh, Lf)) / (10.0 * len(Q))
    print(f"selectivity {sel}: {len(Q)/dt:.0f} QPS, recall@10 {rec:.4f}, n_eval/q {st.last_n_eval/len(Q):.0f}, hops/q {st.last_n_hops/len(Q):.0f}", flush=True)
