"""What carrying a cancellation word costs a batch that is never cancelled: FLAT 10M x 768 B=256 and HNSW 10M-row-free variant
(2M rows) through the host entry point, with cancel=None and with a flag that stays down (the kernels then poll a word in pinned
host memory; the waiting thread polls the stream instead of sleeping in hipStreamSynchronize)."""
import ctypes as C, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows, device_view
dev = torch.device("cuda", 0)
N, D, B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768, 256, 10
ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
p, stride = ix.device_rows(N)
t = device_view(p, (N, stride // 4), dev)
for lo, x in gen_rows(0, N, D, dev):
    t[lo:lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N)
g = torch.Generator(device=dev); g.manual_seed(7)
Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1).cpu().numpy()
flag = C.c_int(0)
for name, kw in (("no flag", {}), ("flag down", {"cancel": flag}), ("no flag", {}), ("flag down", {"cancel": flag})):
    ix.search_batch(Q, K, **kw)
    t0 = time.perf_counter()
    for _ in range(20):
        ix.search_batch(Q, K, **kw)
    print(f"FLAT {N}x{D} B={B} host entry, {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per batch", flush=True)
