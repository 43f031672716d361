import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import _pkg
vsa = _pkg.vsa
n, dim = 4_000_000, 768
x = np.random.default_rng(0).standard_normal((n, dim), dtype=np.float32)
g = vsa.Index("FLAT", dim, "L2", initial_cap=n)
t0 = time.perf_counter(); g.add_batch(x); g.flush(); dt = time.perf_counter() - t0
print(f"FLAT add_batch {n}x{dim}: {dt:.2f} s = {n/dt/1e6:.2f} M rows/s = {n*dim*4/dt/1e9:.1f} GB/s")
