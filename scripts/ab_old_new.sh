for lib in OLD NEW; do
  if [ $lib = OLD ]; then export VKINDEX_LIB=$PWD/build/old_r02/libvkindex.so; else unset VKINDEX_LIB; fi
  python bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --steps 20 --single-query-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; b=j['config3_flat_bf16_ip']
print('$lib', 'f32 kernel', r['per_launch_ms'], 'step', r['step_ms_on_stream'], '| bf16 step', b['ms_per_step'], b['parity_vs_oracle'])"
done
