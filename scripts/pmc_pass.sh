#!/bin/bash
# (VK_GEMM_ABLATE / VK_GEMM_MODE are read by the -DVK_EXPERIMENTS build of the library only)
make -s -j8 -C "$(dirname "$0")/../valkey-search_amd/csrc" experiments && export VKINDEX_LIB="$(cd "$(dirname "$0")/.." && pwd)/valkey-search_amd/libvkindex_exp.so"
# One rocprofv3 --pmc pass over a short bench run.  Usage: scripts/pmc_pass.sh TAG "COUNTER ..." [mode:lockstep]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CNT=$2; cfg=${3:-0:1}
M=${cfg%%:*}; W=${cfg##*:}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$ROOT/gpurun_out/pmc_$TAG
rm -rf $D
VK_GEMM_MODE=$M VK_GEMM_LOCKSTEP=$W timeout 300 rocprofv3 --pmc $CNT -d $D --output-format csv -- \
  python $ROOT/bench.py ${BENCH_EXTRA:-} --steps 2 --warmup 1 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 > $D.log 2>&1
echo "== $TAG mode $M lockstep $W (rc $?)"
python $ROOT/scripts/pmc_agg.py $D > $ROOT/gpurun_out/pmc_$TAG.json
python - $ROOT/gpurun_out/pmc_$TAG.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j.items():
    if "flat_gemm" in k:
        print(k, json.dumps(v))
PY
tail -3 $D.log | cut -c1-300
rm -rf $D
