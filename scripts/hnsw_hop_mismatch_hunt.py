"""Hunt for a graph on which the device's hop counter differs from the oracle's (seen once in nine runs of
tests/test_hnsw_visited_hash_gpu.py at ef = 544: 38092 vs 38093 hops over 70 queries, ids and distances identical).
Graphs built by the host builder with 4 threads differ from build to build; searches on a fixed graph are deterministic
(scripts/hnsw_counter_stress.py)."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import _pkg
vsa = _pkg.vsa
from oracle import oracle
os.environ["VK_HNSW_VISITED_HASH"] = "2"
n, dim, M = 5000, 48, 8
found = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    rng = np.random.default_rng(7000 + seed)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=4)
    g.add_batch(x)
    g.flush()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    for ef, vm in [(e, m) for m in (1, 0, 2, 3) for e in (512, 544, 700, 300)]:
        g.set_option("hnsw-visited-mode", vm)
        D, L, N = g.search_batch(Q, 10, ef=ef)
        st = g.stats()
        per = [o.search(Q[i], 10, ef=ef, stats=True) for i in range(len(Q))]
        ne, nh = sum(p[2] for p in per), sum(p[3] for p in per)
        if (st.last_n_eval, st.last_n_hops) != (ne, nh):
            found += 1
            print(f"seed {seed} ef {ef} mode {vm}: device ({st.last_n_eval}, {st.last_n_hops}) oracle ({ne}, {nh}) kernel mode {st.last_visited_mode}", flush=True)
            for i in range(len(Q)):
                g.search_batch(np.repeat(Q[i:i + 1], 8, axis=0), 10, ef=ef)
                s1 = g.stats()
                if (s1.last_n_eval, s1.last_n_hops) != (8 * per[i][2], 8 * per[i][3]):
                    od, ol = per[i][0], per[i][1]
                    print(f"   query {i}: device x8 ({s1.last_n_eval}, {s1.last_n_hops}) oracle ({per[i][2]}, {per[i][3]}); oracle's 10th / worst kept distance bits {od[-1].view(np.uint32):#x}", flush=True)
                    for mode in (0, 1, 2, 4):
                        g.set_option("hnsw-visited-mode", mode)
                        g.search_batch(np.repeat(Q[i:i + 1], 8, axis=0), 10, ef=ef)
                        s2 = g.stats()
                        print(f"      mode {mode} (kernel {s2.last_visited_mode}): ({s2.last_n_eval // 8}, {s2.last_n_hops // 8})", flush=True)
                    g.set_option("hnsw-visited-mode", vm)
                    d1, l1 = g.search(Q[i], 10, ef=ef)
                    s3 = g.stats()
                    print(f"      one query alone (latency kernel): ({s3.last_n_eval}, {s3.last_n_hops})", flush=True)
    del g
print("graphs with a mismatch:", found)
