import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import _pkg
vsa = _pkg.vsa
from oracle import oracle
rng = np.random.default_rng(5)
n, dim, M = 30000, 48, 32
x = rng.standard_normal((n, dim)).astype(np.float32)
os.environ["VK_HNSW_VISITED_HASH"] = "2"
g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=100, build_threads=8)
g.add_batch(x); g.flush()
o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=100)
Q = rng.standard_normal((200, dim)).astype(np.float32)
for mode in (0, 4):
    g.set_option("hnsw-visited-mode", mode)
    for ef in (128, 160):
        D, L, N = g.search_batch(Q, 10, ef=ef)
        st = g.stats()
        bad = 0; ne = nh = 0
        for i in range(len(Q)):
            od, ol, e, h = o.search(Q[i], 10, ef=ef, stats=True)
            ne += e; nh += h
            if L[i, :N[i]].tolist() != ol.tolist() or D[i, :N[i]].view(np.uint32).tolist() != od.view(np.uint32).tolist(): bad += 1
        print(f"mode {mode} ef {ef}: wrong answers {bad}/{len(Q)}, redo {st.last_frontier_redo}, evals {st.last_n_eval} vs oracle {ne}, hops {st.last_n_hops} vs {nh}", flush=True)
