for p in 262144 131072 65536 32768; do
  echo "== VK_FILTER_PREPASS=$p"
  VK_FILTER_PREPASS=$p python bench.py --steps 30 --warmup 5 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 --no-cpu-baseline --no-serving 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline'])
x=j.get('extras',{}) or j
for k in ('filter','flat_filter','candidates'):
    if k in x: print(k, x[k])
"
done
