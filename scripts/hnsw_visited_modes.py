"""10M x 768 HNSW (configs[2]), 8192 queries: the batch kernel with the three ways of keeping the visited set
(option hnsw-visited-mode: 0 = compare-and-swap at agent scope, 1 = at wavefront scope, 2 = buckets with their fill
counts in LDS, no atomics on memory) on ONE graph, answers and work counters compared between the modes."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows
dev = torch.device("cuda", 0)
N, D, nq = int(os.environ.get("ROWS", 10_000_000)), 768, int(os.environ.get("NQ", 8192))
efs = [int(e) for e in os.environ.get("EFS", "128,768").split(",")]
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(nq, D, generator=g, device=dev), dim=1).cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=128)
step = 1_000_000
t = time.time()
for lo in range(0, N, step):
    x = torch.empty(min(step, N - lo), D, device=dev)
    for l2, c in gen_rows(lo, x.shape[0], D, dev):
        x[l2 - lo:l2 - lo + c.shape[0]] = c
    h.add_batch(x.cpu().numpy(), np.arange(lo, lo + x.shape[0], dtype=np.uint64))
h.flush()
print(f"built {N} x {D} in {time.time() - t:.1f} s", flush=True)
h.set_option("hnsw-visited-hash", 2)
modes = [int(m) for m in os.environ.get("MODES", "0,1,2,0,2").split(",")]
for ef in efs:
    ref = None
    for mode in modes:
        h.set_option("hnsw-visited-mode", mode)
        h.search_batch(Q, 10, ef=ef)
        t = time.time()
        reps = 5
        for _ in range(reps):
            Dh, Lh, Nh = h.search_batch(Q, 10, ef=ef)
        dt = (time.time() - t) / reps
        st = h.stats()
        useful = st.last_n_eval * (D * 4 + 4) + st.last_n_hops * 132
        cur = (Dh.view(np.uint32).copy(), Lh.copy(), st.last_n_eval, st.last_n_hops)
        same = "-" if ref is None else str(bool((cur[0] == ref[0]).all() and (cur[1] == ref[1]).all() and cur[2:] == ref[2:]))
        if ref is None: ref = cur
        print(f"ef={ef} mode {mode}: {nq/dt:.0f} QPS ({dt*1e3:.2f} ms per batch incl. host copies), useful {useful/dt/1e12:.2f} TB/s, "
              f"evals/q {st.last_n_eval/nq:.0f} hops/q {st.last_n_hops/nq:.0f} redo {st.last_frontier_redo}, same as mode {modes[0]}: {same}", flush=True)
