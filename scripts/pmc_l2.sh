#!/bin/bash
# rocprofv3 --pmc passes over the batched L2 scan (K3, 8 queries per pass) at 2M x 768, B=256.
# Usage: scripts/pmc_l2.sh "ENV=VAL ..." -- each counter group is its own pass
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/l2run.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["VKROOT"])
import _pkg
vsa = _pkg.vsa
N, D = 2_000_000, 768
rng = np.random.default_rng(1)
ix = vsa.Index("FLAT", D, "L2", initial_cap=N)
for lo in range(0, N, 250_000):
    ix.add_batch(rng.standard_normal((250_000, D), dtype=np.float32), np.arange(lo, lo + 250_000, dtype=np.uint64))
Q = rng.standard_normal((256, D), dtype=np.float32)
for _ in range(3): ix.search_batch(Q, 10)
PY
i=0
for CNT in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1)); Dd=$ROOT/gpurun_out/pmc_l2_$i; rm -rf $Dd
  env $1 VKROOT=$ROOT timeout 300 rocprofv3 --pmc $CNT -d $Dd --output-format csv -- python /tmp/l2run.py > $Dd.log 2>&1
  python $ROOT/scripts/pmc_agg.py $Dd | python -c "
import json,sys
j=json.load(sys.stdin)
for k,v in j.items():
    if 'flat_scan' in k: print(k[:60], json.dumps(v))
"
  rm -rf $Dd
done
