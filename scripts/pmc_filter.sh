#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run) over the candidate-filter kernel of a short bench run.
# Usage: scripts/pmc_filter.sh ["ENV=VAL ..."]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); Dd=$ROOT/gpurun_out/pmc_filter_$i; rm -rf $Dd
  env $1 timeout 300 rocprofv3 --pmc $CNT -d $Dd --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $Dd.log 2>&1
  python $ROOT/scripts/pmc_agg.py $Dd | python -c "
import json,sys
j=json.load(sys.stdin)
for k,v in j.items():
    if 'flat_filter' in k: print(json.dumps(v))
"
  rm -rf $Dd
done
