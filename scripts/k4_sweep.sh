#!/bin/bash
# K4 (MFMA FLAT) A/B runs on one box: lockstep window sweep.  Usage: scripts/k4_sweep.sh "0 1 2"
mkdir -p gpurun_out
for W in ${1:-0 1 2}; do
  echo "== VK_GEMM_LOCKSTEP=$W"
  VK_GEMM_LOCKSTEP=$W timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 2>&1 | grep -v amdgpu.ids | tail -1
done
