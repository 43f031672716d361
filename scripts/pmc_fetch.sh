#!/bin/bash
# (VK_GEMM_ABLATE / VK_GEMM_MODE are read by the -DVK_EXPERIMENTS build of the library only)
make -s -j8 -C "$(dirname "$0")/../valkey-search_amd/csrc" experiments && export VKINDEX_LIB="$(cd "$(dirname "$0")/.." && pwd)/valkey-search_amd/libvkindex_exp.so"
# HBM-side traffic of the bench kernels: one rocprofv3 --pmc FETCH_SIZE pass per K4 configuration.
# Usage: scripts/pmc_fetch.sh "mode:lockstep ..."
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in ${1:-0:0 2:1}; do
  M=${cfg%%:*}; W=${cfg##*:}
  D=$ROOT/gpurun_out/pmc_m${M}_w${W}
  rm -rf $D
  VK_GEMM_MODE=$M VK_GEMM_LOCKSTEP=$W timeout 420 rocprofv3 --pmc FETCH_SIZE -d $D --output-format csv -- \
    python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-query-steps 1 --hnsw-rows 0 > $D.log 2>&1
  echo "== mode $M lockstep $W (rc $?)"
  python $ROOT/scripts/pmc_agg.py $D | tee $ROOT/gpurun_out/pmc_m${M}_w${W}.json
  rm -rf $D
done
