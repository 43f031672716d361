"""FLAT 10M x 768, B = 256: K steps enqueued on ONE stream against the same K steps alternating between TWO streams (two
batches in flight, as the dispatcher keeps them): what the small launches around the final pass cost when another batch's
final pass can run beside them."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _pkg
vsa = _pkg.vsa
from bench import gen_rows, device_view_typed, make_queries
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
D, B, K, STEPS = 768, 256, 10, 20
ix = vsa.Index("FLAT", D, "COSINE" if dtype == "f32" else "IP", initial_cap=N, dtype=dtype)
ptr, stride = ix.device_rows(N)
if dtype == "f32":
    table = device_view_typed(ptr, (N, stride // 4), dev, "<f4")
else:
    table = device_view_typed(ptr, (N, stride // 2), dev, "<i2").view(torch.bfloat16)
for lo, x in gen_rows(0, N, D, dev):
    table[lo: lo + x.shape[0], :D] = x
torch.cuda.synchronize()
ix.commit_device_rows(N, np.arange(N, dtype=np.uint64))
gA = torch.Generator(device=dev); gA.manual_seed(1234)
A = torch.randn(D, 32, generator=gA, device=dev)
Qs = [make_queries(A, B, D, dev, 4242 + i) for i in range(2)]
outs = [(torch.empty(B, K, device=dev), torch.empty(B, K, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def run(nstreams):
    def step(i):
        s = i % nstreams
        od, ol, on = outs[i % 2]
        ix.search_batch_device(Qs[i % 2].data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=streams[s].cuda_stream)
    for i in range(4): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(STEPS): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / STEPS * 1e3
ref = None
for rep in range(3):
    a = run(1); ans1 = [o[1].clone() for o in outs]
    b = run(2); ans2 = [o[1].clone() for o in outs]
    same = all((x == y).all().item() for x, y in zip(ans1, ans2))
    print(f"rows {N} {dtype}: one stream {a:.3f} ms/step = {B / a * 1e3:.0f} QPS | two streams {b:.3f} ms/step = {B / b * 1e3:.0f} QPS ({(a - b) * 1e3:.0f} us less), answers identical: {same}", flush=True)
