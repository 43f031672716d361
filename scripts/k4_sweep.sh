#!/bin/bash
# (VK_GEMM_ABLATE / VK_GEMM_MODE are read by the -DVK_EXPERIMENTS build of the library only)
make -s -j8 -C "$(dirname "$0")/../valkey-search_amd/csrc" experiments && export VKINDEX_LIB="$(cd "$(dirname "$0")/.." && pwd)/valkey-search_amd/libvkindex_exp.so"
# K4 (MFMA FLAT) A/B runs on one box.  Usage: scripts/k4_sweep.sh "mode:lockstep ..."   e.g. "0:0 0:1 2:1"
mkdir -p gpurun_out
for cfg in ${1:-0:0 0:1 2:1}; do
  M=${cfg%%:*}; W=${cfg##*:}
  echo "== VK_GEMM_MODE=$M VK_GEMM_LOCKSTEP=$W"
  VK_GEMM_MODE=$M VK_GEMM_LOCKSTEP=$W timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        j = json.loads(l); print('   ms_per_step', j['ms_per_step'], 'qps', round(j['value'], 1), 'roofline', j['roofline']['achieved'], j['roofline']['frac'], 'parity', j.get('parity'))
    except Exception as e:
        print('   ??', l.strip()[:300])
"
done
