#!/bin/bash
# Memory-side traffic of the HNSW search kernel against the useful bytes the kernel counts itself (n_eval rows +
# n_hops link lists): is "0.75 of the HBM peak in useful bytes" bandwidth from memory, or hub nodes served from a cache?
#   pass 1  FETCH_SIZE (= L2 -> fabric read requests; Infinity-Cache hits are NOT excluded, MI355X_MICROARCH.md) of the search
#           launches, calibrated on the single-query FLAT scan of the same run (reads each of its rows x 3072 B exactly once with
#           the same 16 B / lane loads; carries the gfx950 1/2 factor)
#   pass 2  TCC_HIT_sum / TCC_MISS_sum: the L2's own hit rate
#   pass 3  whatever memory-side counters the tool lists for DRAM / MALL (printed; summed if collectable)
# Usage: scripts/pmc_hnsw_traffic.sh ROWS NQ EF   (GPU box)  -> gpurun_out/r06_pmc_hnsw_traffic.json + .log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${1:-10000000}; NQ=${2:-8192}; EF=${3:-128}
OUT=$ROOT/gpurun_out/r06_pmc_hnsw_traffic
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT.log
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC_[A-Z0-9_]*(DRAM|MALL|EA0_RDREQ|EA_RDREQ)[A-Z0-9_]*|FETCH_SIZE|MALL[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ' > $OUT.counters
echo "memory-side counters listed by the tool: $(cat $OUT.counters)" >> $OUT.log
i=0
for CNT in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1)); D=$ROOT/gpurun_out/pmc_hnsw_t$i; rm -rf $D
  timeout -s KILL 900 rocprofv3 --pmc $CNT --kernel-include-regex "hnsw_search|flat_scan_kernel<1," -d $D --output-format csv -- \
    python $ROOT/scripts/hnsw_probe.py --rows $ROWS --nq $NQ --ef $EF --calibrate > $D.log 2>&1
  echo "pass $i ($CNT) rc $?" >> $OUT.log; grep -E "^ef=|calibration" $D.log >> $OUT.log
  python $ROOT/scripts/pmc_agg.py $D > $D.agg.json 2>/dev/null
  rm -rf $D
done
python - $OUT $ROWS $NQ $EF <<'PY'
import json, re, sys
out, rows, nq, ef = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import source_sha256
res = {"rows": rows, "queries_per_launch": nq, "ef": ef, "src_sha256": source_sha256(),
       "note": "rocprofv3 --pmc, one pass per counter group, search launches of scripts/hnsw_probe.py only; FETCH_SIZE calibrated on the "
               "single-query FLAT scan of the same run (rows x 3072 B read exactly once)"}
def agg(i):
    try:
        return json.load(open(out.replace("r06_pmc_hnsw_traffic", "pmc_hnsw_t%d" % i) + ".agg.json"))
    except Exception:
        return {}
log = open(out + ".log").read()
m = re.search(r"^ef=%d: .*n_eval/q=(\d+) hops/q=(\d+), useful (\d+) GB/s" % ef, log, re.M)
if m:
    n_eval, hops = int(m.group(1)), int(m.group(2))
    res["useful_bytes_per_launch"] = nq * (n_eval * (768 * 4 + 4) + hops * 132)
    res["useful_gbs"] = int(m.group(3))
a1 = agg(1)
cal = [k for k in a1 if "flat_scan_kernel<1," in k and "[small]" not in k]
srch = [k for k in a1 if "hnsw_search" in k and "[small]" not in k]
if cal and srch:
    c = max(a1[k]["FETCH_SIZE"] for k in cal)
    res["calibration"] = {"kernel": cal[0], "fetch_size": c, "bytes": rows * 3072.0}
    res["kernels"] = {k: {"fetch_size": a1[k]["FETCH_SIZE"], "dispatches": a1[k]["dispatches"], "bytes_per_launch": a1[k]["FETCH_SIZE"] * rows * 3072.0 / c} for k in srch}
    # the same kernel template also serves the BUILD's beam searches (hundreds of launches at efConstruction): the timed loop is
    # the instantiation with exactly the probe's 5 repetitions
    timed = [k for k in srch if a1[k]["dispatches"] == 5]
    top = max(timed or srch, key=lambda k: a1[k]["FETCH_SIZE"])
    res["search_kernel"] = top
    res["traffic_bytes_per_launch"] = res["kernels"][top]["bytes_per_launch"]
    if "useful_bytes_per_launch" in res:
        res["traffic_over_useful"] = round(res["traffic_bytes_per_launch"] / res["useful_bytes_per_launch"], 4)
a2 = agg(2)
for k, v in a2.items():
    if "hnsw_search" in k and "TCC_HIT_sum" in v and "[small]" not in k:
        res.setdefault("l2", {})[k] = {"hit": v["TCC_HIT_sum"], "miss": v["TCC_MISS_sum"], "hit_rate": round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)}
if res.get("search_kernel") in res.get("l2", {}):
    res["l2_hit_rate"] = res["l2"][res["search_kernel"]]["hit_rate"]
for i in (3,):
    for k, v in agg(i).items():
        if "hnsw_search" in k and "[small]" not in k:
            res.setdefault("fabric", {}).setdefault(k, {}).update({c: x for c, x in v.items() if c != "dispatches"})
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $OUT.log
