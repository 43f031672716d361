"""Where the fused re-rank's time goes, stage by stage: the experiments build stamps wall_clock64 (100 MHz) at the stage
boundaries of every query's first block (wave 0, lane 0); this prints the medians over the queries of a batch.
  python scripts/rerank_stamps.py [rows]        (builds and loads valkey-search_amd/libvkindex_exp.so)"""
import os, sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import _pkg
exp = _pkg.vsa.build_experiments() if not os.environ.get("VKINDEX_LIB") else Path(os.environ["VKINDEX_LIB"])
if not os.environ.get("VKINDEX_LIB"):
    os.environ["VKINDEX_LIB"] = str(exp)
    os.execv(sys.executable, [sys.executable] + sys.argv)
import numpy as np, torch
vsa = _pkg.vsa
from bench import gen_rows
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D, B, K = 768, 256, 10
ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
ptr, stride = ix.device_rows(N)
from bench import device_view_typed, make_queries
table = device_view_typed(ptr, (N, stride // 4), dev, "<f4")
for lo, x in gen_rows(0, N, D, dev):
    table[lo: lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
q = make_queries(A, B, D, dev, 4242)
od = torch.empty(B, K, device=dev); ol = torch.empty(B, K, dtype=torch.int64, device=dev); on = torch.empty(B, dtype=torch.int32, device=dev)
stamps = torch.zeros(B, 16, dtype=torch.int64, device=dev)
lib = vsa.lib()
lib.vk_exp_rerank_stamps.argtypes = [C.c_void_p]; lib.vk_exp_rerank_stamps.restype = None
lib.vk_exp_rerank_stamps(stamps.data_ptr())
for _ in range(5):
    ix.search_batch_device(q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
names = ["start", "survivor count", "list + scores", "tile norms", "bound (bisection)", "block barrier", "compaction", "exact distances", "LDS merge", "answer written"]
t0 = s[:, 0].min()
print(f"rows {N}: stage boundaries of the queries' first blocks, wave 0 (10 ns ticks); block starts spread over {(s[:,0].max()-t0)/100:.1f} us")
for i in range(1, 10):
    d = (s[:, i] - s[:, i - 1]) / 100.0
    print(f"  {names[i]:22s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}")
print(f"  first start -> last answer: {(s[:, 9].max() - t0) / 100.0:.1f} us; per block start -> answer median {np.median(s[:,9]-s[:,0])/100:.1f} us")
