"""10M graph: which visited-set arrangement at which ef -- thresholds of the two LDS sets swept through their options."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows
dev = torch.device("cuda", 0)
N, D, nq = 10_000_000, 768, 8192
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(nq, D, generator=g, device=dev), dim=1).cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=128)
for lo in range(0, N, 1_000_000):
    x = torch.empty(1_000_000, D, device=dev)
    for l2, c in gen_rows(lo, x.shape[0], D, dev):
        x[l2 - lo:l2 - lo + c.shape[0]] = c
    h.add_batch(x.cpu().numpy(), np.arange(lo, lo + x.shape[0], dtype=np.uint64))
h.flush()
def run(ef, tag, **opts):
    for k, v in opts.items(): h.set_option(k, v)
    h.search_batch(Q, 10, ef=ef)
    t = time.time()
    for _ in range(4): Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)
    dt = (time.time() - t) / 4
    st = h.stats()
    useful = st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132
    print(f"ef={ef:4d} {tag:28s}: {nq/dt:8.0f} QPS, useful {useful/dt/1e12:.2f} TB/s, evals/q {st.last_n_eval/nq:.0f}, redo {st.last_frontier_redo}", flush=True)
    return (Dh.view(np.uint32).copy(), Lh.copy(), st.last_n_eval, st.last_n_hops)
for ef in (128, 160, 192, 256, 384, 512):
    w = ef * 32
    ref = run(ef, "table in memory (mode 0)", **{"hnsw-visited-mode": 0})
    a = run(ef, "12 KB LDS set + spill", **{"hnsw-visited-mode": 3, "hnsw-lds-visited-work": 1 << 20, "hnsw-lds-visited-work-big": 0})
    b = run(ef, "32 KB LDS set + spill", **{"hnsw-visited-mode": 3, "hnsw-lds-visited-work": 0, "hnsw-lds-visited-work-big": 1 << 20})
    for nm, r in (("12 KB", a), ("32 KB", b)):
        same = (r[0] == ref[0]).all() and (r[1] == ref[1]).all() and r[2:] == ref[2:]
        if not same: print(f"   {nm}: answers or counters differ from mode 0")
