#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's FLAT leg alone (timed region + single-query scan): per-kernel durations
# of one batched step.   scripts/prof_flat.sh <tag> [extra bench args]   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_flat}; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$ROOT/gpurun_out/prof_$TAG
rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 "$@" > $ROOT/gpurun_out/${TAG}_bench.log 2>&1
find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $D
python - $ROOT/gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith("Name") or "vk::" in r[0]:
        print(r[0][:90].ljust(90), r[1].rjust(5), r[3][:12].rjust(12), r[5].rjust(10), r[6].rjust(10))
PY
tail -c 900 $ROOT/gpurun_out/${TAG}_bench.log
