"""Single adds from 16 writer threads, 1M x 768 into an HNSW index, repeated: how many of the staged rows ended up with the
host builder (bulks below 4096 rows) and what the ingest took.  A/B of two builds: VKINDEX_LIB=<other .so> in a fresh process."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(3)
A = rng.standard_normal((32, dim)).astype(np.float32)
x = rng.standard_normal((n, 32)).astype(np.float32) @ A + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
for r in range(reps):
    g = vsa.Index("HNSW", dim, "IP", initial_cap=n, m=16, ef_construction=200, ef_runtime=128)
    t0 = time.perf_counter()
    failed, t_adds = vsa.probe_add_single(g, x, threads=16)
    g.flush()
    t = time.perf_counter() - t0
    st = g.stats()
    print(f"run {r}: {t:.2f} s, failed {failed}, staged {st.staged_adds}, linked on the device {st.staged_adds_device}, "
          f"by the host builder {st.staged_adds - st.staged_adds_device}", flush=True)
    del g
