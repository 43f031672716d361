#!/bin/bash
# (VK_GEMM_ABLATE / VK_GEMM_MODE are read by the -DVK_EXPERIMENTS build of the library only)
make -s -j8 -C "$(dirname "$0")/../valkey-search_amd/csrc" experiments && export VKINDEX_LIB="$(cd "$(dirname "$0")/.." && pwd)/valkey-search_amd/libvkindex_exp.so"
# K4 ablation ladder (VK_GEMM_ABLATE, see flat_gemm.hip): where the time of a stage goes.
for A in ${1:-0 1 2 3 4 5}; do
  echo "== VK_GEMM_ABLATE=$A"
  VK_GEMM_ABLATE=$A VK_GEMM_MODE=${2:-7} VK_GEMM_LOCKSTEP=${3:-0} timeout 600 python bench.py ${BENCH_EXTRA:-} --steps 6 --warmup 2 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        j = json.loads(l); print('   ms_per_step', j['ms_per_step'], 'TF', j['roofline']['achieved'])
    except Exception as e:
        print('   ??', l.strip()[:300])
"
done
