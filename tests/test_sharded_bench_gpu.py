"""The N > 1 paths of bench.py on real kernels, two logical shards on cuda:0.
  single process   `python bench.py --gpus 2 --same-device`: ONE vk_index with n_shards = 2, no launcher -- what the
                   driver's multi-GPU run uses.  The line must carry an HBM roofline (frac <= 1, per GPU), the CPU baseline,
                   the parity of the merged answer against the CPU path, and the legs of configs[3] / configs[4].
  torchrun, lib    the same object driven by rank 0 of two ranks (the launcher contract)
  torchrun, ranks  one process per shard, all-gather (gloo stages it through the host; RCCL on real hardware) + device merge
In every mode the merged answer must be bit-identical to the unsharded index's."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SIZES = ["--rows", "600000", "--steps", "3", "--warmup", "1", "--same-device", "--verify-merge", "--hnsw-rows", "40000",
         "--hybrid-rows", "20000", "--bf16-rows", "300000", "--single-query-steps", "0", "--hnsw-queries", "512", "--hybrid-queries", "512"]


def _lines(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return [l for l in lines if "metric" in l], [l for l in lines if "verify_merge" in l and "metric" not in l]


def test_single_process_two_shards():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *SIZES], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    bench, ver = _lines(r)
    assert len(bench) == 1 and ver and ver[0]["verify_merge"] == "bit-identical"
    b = bench[0]
    assert b["n_gpus"] == 2 and b["config"]["sharding"] == "rows/2" and "n_shards=2" in b["config"]["parallelism"]
    rf = b["roofline"]
    assert rf["bound"] == "hbm" and 0.0 < rf["frac"] <= 1.0 and rf["algorithmic_bytes"] == 300000 * 768 * 4
    assert rf["priced"] == "whole step" and "flat_filter_bdma_kernel" in rf["dominant_kernel"]["kernel"] and rf["dominant_kernel"]["launches_timed"] == 3
    assert len(rf["per_shard_main_pass_ms"]) == 2 and all(m and m > 0 for m in rf["per_shard_main_pass_ms"])
    assert rf["dominant_kernel"]["per_launch_ms"] <= rf["step_ms_on_stream"]
    assert b["config"]["peer_access_matrix"] == [[1]]      # (two logical shards on one device)
    assert rf["host_fanout_enqueue_us_per_step"] is not None
    ga = b["gather"]      # peer copies against the in-library RCCL all-gather, same steps, same answer
    assert ga["answers_identical"] is True and ga["rccl_ranks"] == 1 and ga["lists_per_rank"] == 2 and ga["rccl_gathers"] >= 3
    assert ga["rccl_all_gather_ms_per_step"] > 0 and ga["peer_copies_ms_per_step"] > 0
    assert b["cpu_baseline"] and b["cpu_baseline"]["value"] > 0 and b["cpu_baseline"]["kind"] == "port"
    assert b["config"]["parity_vs_oracle"] == "bit-exact"
    c3 = b["config3_sharded_bf16_ip"]
    assert c3["parity_vs_oracle"] == "bit-exact" and c3["rows"] == 600000 and c3["scaling"] == "weak"
    assert c3["roofline"]["bound"] == "hbm" and c3["roofline"]["frac"] <= 1.0
    c4 = b["config4_sharded_hybrid"]
    assert c4["shards"] == 2 and c4["rows_per_shard"] == 20000 and c4["recall_at_10"] >= 0.9
    h = b["hnsw"]
    assert h["shards"] == 2 and h["rows_per_shard"] == 20000 and h["matched_ef"]["recall_at_10"] >= 0.9
    assert h["single_graph"]["at_ef"]["recall_at_10"] > 0.5
    m = h["matched_recall_of_single_graph_at_ef"]
    assert m is not None and m["recall_at_10"] >= m["target_recall"] and m["ef"] <= 128


@pytest.mark.parametrize("mode,port", [("lib", "29533"), ("ranks", "29534")])
def test_two_ranks_two_shards(mode, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, str(ROOT / "bench.py"), "--multi-gpu", mode, "--gpus", "2", "--rows", "300000", "--steps", "2",
           "--warmup", "1", "--backend", "gloo", "--same-device", "--verify-merge", "--hnsw-rows", "40000", "--hybrid-rows", "0",
           "--bf16-rows", "0", "--no-cpu-baseline", "--no-hnsw-single-ref", "--single-query-steps", "0", "--hnsw-queries", "512"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    bench, ver = _lines(r)
    assert len(bench) == 1 and bench[0]["n_gpus"] == 2 and bench[0]["config"]["sharding"] == "rows/2"
    assert ("n_shards=2" in bench[0]["config"].get("parallelism", "")) == (mode == "lib")
    assert ver and ver[0]["verify_merge"] == "bit-identical"
    h = bench[0]["hnsw"]
    rec = h["matched_ef"]["recall_at_10"] if mode == "lib" else h["recall_at_10"]
    assert h["shards"] == 2 and h["rows_per_shard"] == 20000 and rec >= 0.9
