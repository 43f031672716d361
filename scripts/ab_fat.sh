for fat in 0 1; do
  VK_FILTER_FAT=$fat python bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --steps 20 --single-query-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; b=j['config3_flat_bf16_ip']
print('FAT=$fat', 'f32 kernel', r['per_launch_ms'], 'frac', r['frac'], 'step', r['step_ms_on_stream'], r['filter'], '| bf16 step', b['ms_per_step'], b['parity_vs_oracle'])"
done
