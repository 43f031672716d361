"""Are a search's answers and work counters the same every time?  One small graph (5000 x 48, M = 8: the shape of
tests/test_hnsw_visited_hash_gpu.py), every visited-set mode, ef 300 / 512 / 544 / 700, the same 70 queries 200 times each."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
rng = np.random.default_rng(1545)
n, dim, M = 5000, 48, 8
x = rng.standard_normal((n, dim)).astype(np.float32)
os.environ["VK_HNSW_VISITED_HASH"] = "2"
g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=4)
g.add_batch(x)
g.flush()
Q = rng.standard_normal((70, dim)).astype(np.float32)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for mode in (0, 1, 2, 3, 4):
    g.set_option("hnsw-visited-mode", mode)
    for ef in (300, 512, 544, 700):
        ref, bad = None, []
        for r in range(reps):
            D, L, N = g.search_batch(Q, 10, ef=ef)
            st = g.stats()
            cur = (L.tobytes(), D.tobytes(), int(st.last_n_eval), int(st.last_n_hops), int(st.last_frontier_redo))
            if ref is None:
                ref = cur
            elif cur != ref:
                bad.append((r, cur[0] == ref[0], cur[1] == ref[1], cur[2] - ref[2], cur[3] - ref[3], cur[4] - ref[4]))
        print(f"mode {mode} ef {ef}: kernel mode {g.stats().last_visited_mode}, evals {ref[2]} hops {ref[3]} redo {ref[4]}; {len(bad)} of {reps - 1} repeats differ {bad[:4]}", flush=True)
