#!/usr/bin/env python3
"""How fast can random 3 KB rows be gathered from a 30 GB table on this GPU?  (context for the HNSW kernels'
'useful GB/s': torch.index_select of random rows, with the sequential write of the result subtracted)"""
import sys, time, torch
dev = torch.device("cuda", 0)
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768
x = torch.empty(N, D, device=dev)
x.normal_()
M = min(2_000_000, N)
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randint(0, N, (M,), device=dev, generator=g)
out = torch.empty(M, D, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tg = t(lambda: torch.index_select(x, 0, idx, out=out))
seq = x[:M]
tc = t(lambda: out.copy_(seq))
b = M * D * 4
print(f"rows={N}: gather {tg*1e3:.2f} ms ({b/tg/1e9:.0f} GB/s read + same written), sequential copy {tc*1e3:.2f} ms ({b/tc/1e9:.0f} GB/s each way)")
print(f"  random-read rate if the write costs what it costs in the copy: {b/max(tg - tc/2, 1e-9)/1e9:.0f} GB/s")
# read-only variant: embedding_bag sums bags of 32 random rows (the output is 1/32 of the bytes read)
import torch.nn.functional as F
M2 = 4_000_000
idx2 = torch.randint(0, N, (M2,), device=dev, generator=g)
offs = torch.arange(0, M2, 32, device=dev)
te = t(lambda: F.embedding_bag(idx2, x, offs, mode="sum"))
print(f"  read-only gather (embedding_bag, bags of 32): {te*1e3:.2f} ms = {M2*D*4/te/1e9:.0f} GB/s")
