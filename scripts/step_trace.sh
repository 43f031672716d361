#!/bin/bash
# Where a batched FLAT step's time goes, launch by launch: rocprofv3 --kernel-trace of bench.py's FLAT leg, then for the LAST
# step of the timed region every kernel with its start offset, duration and the idle gap in front of it.
#   scripts/step_trace.sh <tag> [bench args, e.g. --rows 1250000]   -> gpurun_out/<tag>_step_trace.log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-step}; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=$ROOT/gpurun_out/trace_$TAG
rm -rf $D
timeout -s KILL 420 rocprofv3 --kernel-trace -d $D --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 --single-query-steps 0 --no-serving --steps 10 --warmup 3 "$@" > $ROOT/gpurun_out/${TAG}_bench.log 2>&1
python - $D <<'PY' | tee $ROOT/gpurun_out/${TAG}_step_trace.log
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vk::", "")))
rows.sort()
# a step starts with flat_qprep_kernel; take the last complete one before the end of the timed region (= the 10th from the end
# of the qprep launches would be a warm-up; the serving legs are off, so the last qprep opens the last timed step)
starts = [i for i, r in enumerate(rows) if r[2].startswith("flat_qprep_kernel")]
if len(starts) < 3:
    print("no batched steps in the trace"); sys.exit(0)
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
def show(lo, hi):
    t0 = rows[lo][0]; prev_end = None; busy = 0
    for s, e, n in rows[lo:hi]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"  +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {n[:70]}")
        prev_end = e if prev_end is None else max(prev_end, e); busy += e - s
    total = (rows[hi][0] - t0) / 1e3
    print(f"  step (qprep start -> next qprep start) {total:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {total - busy / 1e3:.1f} us")
# bench.py: 3 warm-up steps, the 10 TIMED steps (no event records), then 2 + 10 steps with the library's event pairs around the
# main pass (the dominant-kernel figure): the timed region is steps 3 .. 12
timed = steps[3:13] if len(steps) >= 13 else steps
print("last but one step of the timed region:")
show(*timed[-2])
import statistics
tot = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in timed[-9:]]
print("last 9 step periods of the timed region (us):", [round(t, 1) for t in tot], "median", round(statistics.median(tot), 1))
if len(steps) >= 25:
    print("a step of the second region (kernel-timing on: an event record in front of and behind the main pass):")
    show(*steps[-3])
PY
tail -c 600 $ROOT/gpurun_out/${TAG}_bench.log
rm -rf $D
