#!/bin/bash
# Counters of the HNSW search kernel only (--kernel-include-regex keeps the build's hundreds of launches out of the
# counter collection; every group is its own pass).  Usage: pmc_hnsw.sh ROWS NQ [EXTRA_COUNTER_GROUP] [KERNEL_REGEX]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${1:-2000000}; NQ=${2:-8192}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "$3"; do
  [ -z "$CNT" ] && continue
  i=$((i+1)); D=$ROOT/gpurun_out/pmc_hnsw_$i; rm -rf $D
  timeout -s KILL 240 rocprofv3 --pmc $CNT --kernel-include-regex "${4:-hnsw_search_kernel}" -d $D --output-format csv -- python $ROOT/scripts/hnsw_probe.py --rows $ROWS --nq $NQ > $D.log 2>&1
  echo "pass $i rc $?"; grep "ef=" $D.log
  python - $D <<'PY'
import csv, glob, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(float)); names = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
for d in sorted(rows, key=int)[-4:]:
    print(d, names[d][:60], {k: f"{v:.4g}" for k, v in rows[d].items()})
PY
  rm -rf $D
done
