"""seed 40 of scripts/hnsw_hop_mismatch_hunt.py: is the graph on the device the graph that was saved?"""
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
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(7000 + seed)
x = rng.standard_normal((n, dim)).astype(np.float32)
for attempt in range(40):
    g = vsa.Index("HNSW", dim, "L2", initial_cap=n, m=M, ef_construction=40, build_threads=threads)
    g.add_batch(x)
    g.flush()
    chunks = g.save()
    o = oracle.HNSW.from_product_index(g.save_raw, dim, "L2", M, ef_construction=40)
    if attempt == 0:
        Q0 = rng.standard_normal((70, dim)).astype(np.float32)
    Q = Q0
    ef = 300
    D, L, N = g.search_batch(Q, 10, ef=ef)
    st = g.stats()
    per = [o.search(Q[i], 10, ef=ef, stats=True) for i in range(len(Q))]
    ne, nh = sum(p[2] for p in per), sum(p[3] for p in per)
    if (st.last_n_eval, st.last_n_hops) == (ne, nh):
        continue
    print(f"attempt {attempt}: device ({st.last_n_eval}, {st.last_n_hops}) oracle ({ne}, {nh})")
    g2 = vsa.Index.load(chunks, "HNSW", dim, "L2", m=M, ef_construction=40, initial_cap=n)
    if g2 is not None:
        g2.search_batch(Q, 10, ef=ef)
        s2 = g2.stats()
        print(f"   the saved stream loaded into a fresh index: ({s2.last_n_eval}, {s2.last_n_hops})")
    chunks2 = g.save()
    print("   saved twice, same bytes:", [bytes(a) for a in chunks] == [bytes(b) for b in chunks2])
    break
else:
    print("no mismatch in 40 builds of this seed")
