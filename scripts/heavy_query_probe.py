"""tests/test_flat_filter_gpu.py::test_a_heavy_query_costs_the_batch_little, with the statistics of both settings of filter-two-pass"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
rng = np.random.default_rng(31)
n, dim = 1_000_000, 128
unit = lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
centres = rng.standard_normal((200, dim)).astype(np.float32)
x = np.empty((n, dim), np.float32)
for i in range(0, n, 100_000):
    x[i:i + 100_000] = unit(centres[rng.integers(0, 200, 100_000)] + 0.5 * rng.standard_normal((100_000, dim)).astype(np.float32))
dup = x[123].copy()
x[300_000:340_000] = dup
f = vsa.Index("FLAT", dim, "COSINE", initial_cap=n, options={"filter-prepass-rows": 1024, "filter-min-rows": 32768, "kernel-timing": 1})
f.add_batch(x)
Q = unit(centres[rng.integers(0, 200, 256)] + 0.5 * rng.standard_normal((256, dim)).astype(np.float32))
Qh = Q.copy()
Qh[77] = dup
for tp in (1, 0, 1, 0):
    f.set_option("filter-two-pass", tp)
    for name, q in (("clean", Q), ("heavy", Qh)):
        f.search_batch(q, 10)
        s0 = f.stats()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            f.search_batch(q, 10)
            ts.append(time.perf_counter() - t0)
        s1 = f.stats()
        kb = s1.filter_batches - s0.filter_batches
        print(f"two-pass {tp} {name}: median {np.median(ts) * 1e3:.3f} ms  main-pass kernel {(s1.filter_kernel_ns - s0.filter_kernel_ns) / 1e3 / max(kb, 1):.1f} us  "
              f"survivors {s1.last_filter_candidates}  reranked {s1.last_filter_reranked}  fallback {s1.last_filter_fallback}  main rows {s1.last_filter_final_rows}", flush=True)
