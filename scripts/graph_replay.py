"""Does a HIP graph buy anything?  One step of the batched FLAT search (FLAT 10M x 768, B = 256: all the launches and memsets
one vk_index_search_batch_device enqueues) captured into a graph on a side stream and replayed, against the same calls
enqueued one by one.  (The answer the graph writes is compared with the plain call's.)"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows, device_view_typed, make_queries
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D, B, K, STEPS = 768, 256, 10, 20
ix = vsa.Index("FLAT", D, "COSINE", initial_cap=N)
ptr, stride = ix.device_rows(N)
table = device_view_typed(ptr, (N, stride // 4), dev, "<f4")
for lo, x in gen_rows(0, N, D, dev):
    table[lo: lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Q = make_queries(A, B, D, dev, 4242)
od, ol, on = torch.empty(B, K, device=dev), torch.empty(B, K, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
ws = torch.cuda.Stream(device=dev)
def call():
    ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=ws.cuda_stream)
for _ in range(4): call()
torch.cuda.synchronize()
ref = ol.clone()
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(STEPS): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / STEPS * 1e3
plain = [timed(call) for _ in range(3)]
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(ws):
        g.capture_begin()
        call()
        g.capture_end()
    ol.zero_()
    g.replay(); torch.cuda.synchronize()
    same = bool((ol == ref).all())
    with torch.cuda.stream(ws):
        graph = [timed(g.replay) for _ in range(3)]
    print(f"rows {N}: plain calls {min(plain):.3f} ms/step, graph replay {min(graph):.3f} ms/step, answers identical: {same}")
except Exception as e:
    print(f"rows {N}: plain calls {min(plain):.3f} ms/step; capture failed: {type(e).__name__}: {str(e)[:200]}")
