#!/bin/bash
# HBM-side traffic of the timed kernels of bench.py's default workload: ONE rocprofv3 --pmc FETCH_SIZE pass (its own
# run, no tracing, as the guide prescribes), aggregated per kernel, calibrated on the single-query scan kernel (which
# reads each row byte exactly once) and stamped with the hash of the kernel sources it was taken with.
# bench.py prints `roofline.traffic` from the resulting file ONLY while that hash equals the current sources'.
#   scripts/pmc_traffic.sh [out.json]        (run on the GPU box; default out: gpurun_out/r06_pmc_fetch_size.json)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/r06_pmc_fetch_size.json}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT" ;; esac
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$ROOT/gpurun_out/pmc_traffic
rm -rf $D
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $D --output-format csv -- \
  python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-query-steps 2 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $D.log 2>&1
echo "rocprofv3 rc $?"
python $ROOT/scripts/pmc_agg.py $D > $D.agg.json
python - "$D.agg.json" "$OUT" <<'PY'
import json, sys
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import source_sha256
agg = json.load(open(sys.argv[1]))
N, D, = 10_000_000, 768
row_bytes = N * D * 4.0
# (the full single-query scan, not the re-rank instantiation of the same kernel: the one that fetched the most)
cal = sorted([k for k in agg if "flat_scan_kernel<1," in k and "[small]" not in k], key=lambda k: -agg[k].get("FETCH_SIZE", 0.0))
out = {"note": "rocprofv3 --pmc FETCH_SIZE, own pass (scripts/pmc_traffic.sh) of bench.py --steps 3 at 10Mx768 f32, B=256, k=10. "
               "FETCH_SIZE carries the gfx950 1/2 factor for 128-B requests and KiB units (MI355X_MICROARCH.md): it is calibrated "
               "on the single-query scan kernel, which reads each of the 30.72e9 row bytes exactly once.",
       "src_sha256": source_sha256(), "kernels": {}}
if cal:
    c = agg[cal[0]]["FETCH_SIZE"]
    out["calibration_kernel"], out["calibration_fetch_size"], out["calibration_bytes"] = cal[0], c, row_bytes
    for k, v in agg.items():
        if "FETCH_SIZE" in v:
            out["kernels"][k] = {"fetch_size": v["FETCH_SIZE"], "dispatches": v["dispatches"],
                                 "hbm_bytes_per_launch": v["FETCH_SIZE"] * row_bytes / c}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out["kernels"].items()}, indent=1))
PY
rm -rf $D
