#!/bin/bash
# the early pass's share of the rows and the sample behind it, 10M and 1.25M rows x 768 f32 (bench.py's FLAT leg alone)
run() {
  python bench.py --steps 30 --warmup 5 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 --no-cpu-baseline --no-serving --single-query-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('   step %.1f us  main pass %.1f us over %d rows  survivors/query %.1f' % (j['ms_per_step']*1e3, r['per_launch_ms']*1e3, r.get('rows_in_launch', 0), r['filter']['survivors_per_query']))"
}
for rows in 10000000 1250000; do
  for cfg in "0 262144" "30 262144" "80 262144" "120 262144" "0 16384" "40 16384" "0 65536" "80 65536"; do
    set -- $cfg
    echo "rows $rows early permille $1 (0 = auto) sample cap $2"
    VK_FILTER_EARLY_PERMILLE=$1 VK_FILTER_PREPASS=$2 run --rows $rows
  done
done
