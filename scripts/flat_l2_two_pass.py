import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import _pkg
from bench import gen_rows, device_view
vsa = _pkg.vsa
N, D, K = 10_000_000, 768, 10
dev = torch.device("cuda", 0)
ix = vsa.Index("FLAT", D, "L2", initial_cap=N)
p, stride = ix.device_rows(N)
t = device_view(p, (N, stride // 4), dev)
for lo, x in gen_rows(0, N, D, dev):
    t[lo:lo + x.shape[0], :D] = x * (1.0 + 0.5 * torch.rand(x.shape[0], 1, device=dev))   # un-normalised rows: the L2 margin's c2 term at work
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
Q = np.ascontiguousarray(t[:256, :D].cpu().numpy()) + np.float32(0.01)
res = {}
for tp in (1, 0, 1, 0):
    ix.set_option("filter-two-pass", tp)
    ix.search_batch(Q, K)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): r = ix.search_batch(Q, K)
    dt = (time.perf_counter() - t0) / reps
    st = ix.stats()
    print(f"L2 10M x 768 B=256 two-pass {tp}: {dt*1e3:.3f} ms, {256/dt:.0f} QPS, survivors/query {st.last_filter_candidates/256:.1f}, main rows {st.last_filter_final_rows}, handed over {st.last_filter_fallback}", flush=True)
    res[tp] = r
a, b = res[1], res[0]
print("answers identical:", bool((a[1] == b[1]).all() and (a[0].view(np.uint32) == b[0].view(np.uint32)).all()))
