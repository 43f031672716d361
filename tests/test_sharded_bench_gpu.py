"""The N > 1 path of bench.py on real kernels: two ranks share cuda:0 (gloo stages the all-gather through
the host; RCCL is what the driver's multi-GPU run uses), each searches its row shard, the per-shard top-k
lists are merged on the device, and the merged answer must be bit-identical to the unsharded index's."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("mode,port", [("lib", "29533"), ("ranks", "29534")])
def test_two_shards_equal_one_index(mode, port):
    """lib: rank 0 holds ONE vk_index with n_shards = 2 (the product's multi-GPU path, logical shards here);
    ranks: one process per shard, all-gather + device merge (the fallback when rank 0 cannot see every GPU)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, str(ROOT / "bench.py"), "--multi-gpu", mode, "--gpus", "2", "--rows", "300000", "--steps", "2",
           "--warmup", "1", "--backend", "gloo", "--same-device", "--verify-merge", "--hnsw-rows", "40000", "--hnsw-sharded",
           "--single-query-steps", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    bench = [l for l in lines if "metric" in l]
    ver = [l for l in lines if "verify_merge" in l]
    assert len(bench) == 1 and bench[0]["n_gpus"] == 2 and bench[0]["config"]["sharding"] == "rows/2"
    assert ("n_shards=2" in bench[0]["config"].get("parallelism", "")) == (mode == "lib")
    assert ver and ver[0]["verify_merge"] == "bit-identical"
    # one HNSW graph per shard, merged the same way: recall against the exact answer over the same rows
    h = bench[0]["hnsw"]
    assert h["shards"] == 2 and h["rows_per_shard"] == 20000 and h["recall_at_10"] >= 0.9
