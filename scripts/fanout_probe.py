"""Host enqueue time and device step time of a sharded FLAT search against the unsharded index over the same rows,
with S LOGICAL shards on one GPU (every code path of the multi-GPU fan-out but the peer copies).
  python scripts/fanout_probe.py --rows 10000000 --shards 8 [--steps 20]
Prints one JSON line: unsharded step ms, sharded step ms (HIP events on the caller's stream), the host time of the
fan-out per step (vk_index_stats.fanout_enqueue_ns / fanout_calls: enqueue of every shard's work + gather + merge) and
the wall time of the vk_index_search_batch_device call itself.  VK_SHARD_THREADS=0 (set before the run) = the r02
behaviour, one thread enqueueing shard after shard."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from bench import vsa  # noqa: E402


def fill(ix, shard, r0, n, D, dev, bf16):
    ptr, stride = (ix.device_rows(n) if shard is None else ix.shard_device_rows(shard, n))
    esz = 2 if bf16 else 4
    tab = (bench.device_view_typed(ptr, (n, stride // 2), dev, "<i2").view(torch.bfloat16) if bf16
           else bench.device_view(ptr, (n, stride // 4), dev))
    if stride != D * esz:
        tab[:, D:] = 0
    for lo, x in bench.gen_rows(r0, n, D, dev):
        tab[lo - r0: lo - r0 + x.shape[0], :D] = x
    torch.cuda.synchronize()
    labels = np.arange(r0, r0 + n, dtype=np.uint64)
    if shard is None:
        ix.commit_device_rows(n, labels)
    else:
        ix.shard_commit_device_rows(shard, n, labels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--metric", default="COSINE")
    ap.add_argument("--skip-unsharded", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, D, B, K, S = a.rows, a.dim, a.batch, a.k, a.shards
    bf16 = a.dtype == "bf16"
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    A = torch.randn(D, 32, generator=g, device=dev, dtype=torch.float32)
    Q = bench.make_queries(A, B, D, dev, 4242)
    od = torch.empty(B, K, device=dev, dtype=torch.float32)
    ol = torch.empty(B, K, device=dev, dtype=torch.int64)
    on = torch.empty(B, device=dev, dtype=torch.int32)
    ws = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ws)
    out = {"rows": N, "dim": D, "batch": B, "k": K, "shards": S, "dtype": a.dtype, "metric": a.metric,
           "shard_threads": os.environ.get("VK_SHARD_THREADS", "1")}

    def run(ix):
        step = lambda: ix.search_batch_device(Q.data_ptr(), B, K, od.data_ptr(), ol.data_ptr(), on.data_ptr(), stream=ws.cuda_stream)  # noqa: E731
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        st0 = ix.stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call = 0.0
        e0.record()
        for _ in range(a.steps):
            t0 = time.perf_counter()
            step()
            call += time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        st1 = ix.stats()
        calls = st1.fanout_calls - st0.fanout_calls
        return (e0.elapsed_time(e1) / a.steps, call / a.steps * 1e6,
                (st1.fanout_enqueue_ns - st0.fanout_enqueue_ns) / calls / 1e3 if calls else None,
                od.cpu().numpy().copy(), ol.cpu().numpy().copy())

    ref = None
    if not a.skip_unsharded:
        ix = vsa.Index("FLAT", D, a.metric, initial_cap=N, device_id=0, dtype=a.dtype)
        fill(ix, None, 0, N, D, dev, bf16)
        ms, call_us, _, rd, rl = run(ix)
        out["unsharded"] = {"step_ms": round(ms, 4), "call_us": round(call_us, 1)}
        ref = (rd, rl)
        del ix
        torch.cuda.empty_cache()
    ix = vsa.Index("FLAT", D, a.metric, initial_cap=N, dtype=a.dtype, shard_devices=[0] * S)
    for s in range(S):
        r0, r1 = s * N // S, (s + 1) * N // S
        fill(ix, s, r0, r1 - r0, D, dev, bf16)
    ms, call_us, fan_us, sd, sl = run(ix)
    out["sharded"] = {"step_ms": round(ms, 4), "call_us": round(call_us, 1), "fanout_enqueue_us": round(fan_us, 1)}
    if ref is not None:
        out["sharded"]["vs_unsharded"] = round(ms / out["unsharded"]["step_ms"], 4)
        out["sharded"]["bit_identical_to_unsharded"] = bool((ref[0].view(np.uint32) == sd.view(np.uint32)).all() and (ref[1] == sl).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
