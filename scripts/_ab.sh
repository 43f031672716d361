one() { VKINDEX_LIB=$1 bash scripts/step_trace.sh $2 "${@:3}" > /dev/null 2>&1; f=$(grep "flat_filter_b" gpurun_out/$2_step_trace.log | grep -v sample | awk '{print $4}'); m=$(tail -1 gpurun_out/$2_step_trace.log | sed 's/.*median //'); echo "$2: filter kernel $f us, step median $m"; }
for rep in 1 2 3; do
  one $PWD/build_exp/libvkindex_r1.so base_$rep
  one $PWD/valkey-search_amd/libvkindex.so new_$rep
done
for rep in 1 2; do
  one $PWD/build_exp/libvkindex_r1.so basebf_$rep --dtype bf16
  one $PWD/valkey-search_amd/libvkindex.so newbf_$rep --dtype bf16
done
timeout 900 python -m pytest tests/test_flat_filter_gpu.py tests/test_flat_filter_sweep_gpu.py tests/test_flat_filter_adversarial_gpu.py tests/test_bf16_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
