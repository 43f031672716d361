"""World-size-2 `gloo` test (CPU) of the SPECIFICATION the multi-GPU path is built to, not of the product: rows split in
contiguous ranges, per-shard top-k, all-gather of (distance,label)[B][k], merge by (distance,label) -- the merged
answer must be identical to the single-shard answer, exact cross-shard ties included (fanout.cc:162-175
is the reference's cluster-level analogue, whose arrival-order tie rule this total order replaces).  Everything here
is the oracle's (there is no GPU in this container): it pins the rule and the world-size-2 plumbing of bench.py's
fallback mode.  The PRODUCT's sharding, gather and merge kernel are tested where a GPU is:
tests/test_sharded_index_gpu.py (the in-library sharded index against the unsharded one and the oracle) and
tests/test_sharded_bench_gpu.py (both bench.py modes) -- and, without any GPU, in tests/test_host_sanitizers.py: the product's
sharded_index.cc itself over eight virtual devices (the multi-GPU index lives in ONE process, as valkey-server is one: there
is no product code that runs one rank per GPU, so nothing of the product could run under torch.distributed here)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dim, B, k, out_path):
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[n // 2 + 3] = x[5]                      # a cross-shard exact tie: must resolve by label
    Q = rng.standard_normal((B, dim)).astype(np.float32)
    Q[0] = x[5]
    r0, r1 = rank * n // world, (rank + 1) * n // world     # same split as bench.py
    shard = O.Flat(dim, "L2", max_elements=r1 - r0)
    shard.add_many(x[r0:r1], np.arange(r0, r1, dtype=np.uint64))
    d = np.full((B, k), np.inf, np.float32)
    l = np.full((B, k), np.iinfo(np.int64).max, np.int64)
    for i in range(B):
        dd, ll = shard.search(Q[i], k)
        d[i, :len(dd)] = dd
        l[i, :len(ll)] = ll.astype(np.int64)
    all_d = torch.empty(world * B, k, dtype=torch.float32)    # rank-major concatenation == [world][B][k]
    all_l = torch.empty(world * B, k, dtype=torch.int64)
    dist.all_gather_into_tensor(all_d, torch.from_numpy(d))
    dist.all_gather_into_tensor(all_l, torch.from_numpy(l))
    all_d, all_l = all_d.view(world, B, k), all_l.view(world, B, k)
    merged = []
    for i in range(B):
        md, ml = O.merge_topk(all_d[:, i].numpy(), all_l[:, i].numpy().view(np.uint64),
                              np.full(world, k, np.uint32), k)
        merged.append((md, ml))
    if rank == 0:
        full = O.Flat(dim, "L2", max_elements=n)
        full.add_many(x)
        ok = True
        for i in range(B):
            fd, fl = full.search(Q[i], k)
            ok = ok and fl.tolist() == merged[i][1].tolist() and \
                fd.view(np.uint32).tolist() == merged[i][0].view(np.uint32).tolist()
        tie_ok = merged[0][1][:2].tolist() == [5, n // 2 + 3]
        Path(out_path).write_text("ok" if ok and tie_ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_flat_equals_single_shard(tmp_path, world):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), 3001, 24, 7, 5, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
