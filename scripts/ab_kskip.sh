#!/bin/bash
# A/B of the K-step early exit of the final pass (option filter-kskip, env default VK_FILTER_KSKIP) on ONE lease, interleaved:
# the headline step (FLAT 10M x 768 f32 cosine B=256) and the bf16 shard of configs[3], three runs each way.
# usage: scripts/ab_kskip.sh [out]
out=${1:-gpurun_out/r05_ab_kskip.log}
: > "$out"
for rep in 1 2 3; do
  for ks in 1 0; do
    for dt in f32 bf16; do
      line=$(VK_FILTER_KSKIP=$ks python bench.py --dtype $dt --no-cpu-baseline --no-serving --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 --single-query-steps 0 --steps 20 --warmup 3 2>/dev/null | tail -1)
      python - "$ks" "$dt" "$rep" "$line" >> "$out" <<'PY'
import json, sys
ks, dt, rep, line = sys.argv[1:5]
p = json.loads(line)
r = p["roofline"]
print(f"rep {rep} kskip={ks} {dt}: step {p['ms_per_step']:.3f} ms = {p['value']:.0f} QPS, final pass {r['per_launch_ms']:.3f} ms = {r['frac']:.4f} of HBM peak, whole step {r.get('hbm_frac_of_whole_step')}, filter {r.get('filter')}")
PY
    done
  done
done
cat "$out"
