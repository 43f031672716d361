"""10M x 768 HNSW, 8192 queries, ef = 128: the batch kernel with visited tables of 2^13 .. 2^16 words (one graph; a new
index object per setting would rebuild, so the setting is passed through the environment of ONE process per run)."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows
dev = torch.device("cuda", 0)
N, D, nq, ef = int(os.environ.get("ROWS", 10_000_000)), 768, 8192, 128
g = torch.Generator(device=dev); g.manual_seed(4242)
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = torch.nn.functional.normalize(torch.randn(nq, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(nq, D, generator=g, device=dev), dim=1).cpu().numpy()
h = vsa.Index("HNSW", D, "COSINE", initial_cap=N, m=16, ef_construction=200, ef_runtime=ef)
step = 1_000_000
for lo in range(0, N, step):
    x = torch.empty(min(step, N - lo), D, device=dev)
    for l2, c in gen_rows(lo, x.shape[0], D, dev):
        x[l2 - lo:l2 - lo + c.shape[0]] = c
    h.add_batch(x.cpu().numpy(), np.arange(lo, lo + x.shape[0], dtype=np.uint64))
h.flush()
h.search_batch(Q, 10, ef=ef)
t = time.time()
for _ in range(5):
    h.search_batch(Q, 10, ef=ef)
dt = (time.time() - t) / 5
st = h.stats()
print(f"hash log2 {os.environ.get('VK_HNSW_HASH_LOG2', 'default')}: {nq/dt:.0f} QPS, redo {st.last_frontier_redo}, evals/q {st.last_n_eval/nq:.0f}", flush=True)
