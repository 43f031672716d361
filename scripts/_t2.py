import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import _pkg
vsa = _pkg.vsa
M = int(sys.argv[1]); n = int(sys.argv[2]); nq = int(sys.argv[3]); efc = int(sys.argv[4])
rng = np.random.default_rng(5)
dim = 48
x = rng.standard_normal((n, dim)).astype(np.float32)
g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=efc, build_threads=8)
g.add_batch(x); g.flush()
print("built", flush=True)
g.set_option("hnsw-visited-hash", 2)
Q = rng.standard_normal((nq, dim)).astype(np.float32)
for ef in (64, 128, 160):
    D, L, N = g.search_batch(Q, 10, ef=ef)
    st = g.stats()
    print(f"M {M} n {n} nq {nq} ef {ef}: ok, redo {st.last_frontier_redo} evals/q {st.last_n_eval/nq:.0f}", flush=True)
