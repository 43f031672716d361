#!/bin/bash
# Matrix-core utilisation and engine clock of the candidate filter's final pass from counters (own --pmc passes, one per
# row format): SQ_VALU_MFMA_BUSY_CYCLES (= 32 x MFMAs for v_mfma_f32_32x32x16_{f16,bf16}), SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE
# (engine cycles while the kernel ran: / duration = the clock the chip sustained), SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16}.
#   scripts/pmc_mfma.sh            (on the GPU box; writes gpurun_out/r03_pmc_mfma_{f32,bf16}.json)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for DT in f32 bf16; do
  for CNT in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_ANY"; do
    Dd=$ROOT/gpurun_out/pmc_mfma_$DT; rm -rf $Dd
    timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $Dd --output-format csv -- python $ROOT/bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $Dd.log 2>&1
    python $ROOT/scripts/pmc_agg.py $Dd | python -c "
import json,sys
j=json.load(sys.stdin)
print(json.dumps({k:v for k,v in j.items() if 'flat_filter' in k and 'sample' not in k}))"
    python - $Dd <<'PY'
import csv, glob, sys
# average duration of the final-pass kernel in this run (kernel trace of the same pass)
d = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flat_filter" in r["Kernel_Name"] and "sample" not in r["Kernel_Name"]:
            d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if d:
    print({"final_pass_launches": len(d), "avg_ns": sum(d) / len(d)})
PY
    rm -rf $Dd
  done
done 2>&1 | tee $ROOT/gpurun_out/r03_pmc_mfma.log
