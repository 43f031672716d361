#!/bin/bash
# Round-6 evidence, one gpurun call: PMC traffic of the timed kernels (own pass), the default bench line, the rocprofv3
# kernel-stats summaries of the FLAT leg (f32, bf16) and of the 10M HNSW leg, per-launch step timelines, the re-rank's stage
# stamps, the fan-out probe and the 8-logical-shard bench line.  Everything lands in gpurun_out/ (copied to profiles/ by
# hand afterwards).
set -x
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
bash scripts/pmc_traffic.sh 2>&1 | tail -12
cp gpurun_out/r06_pmc_fetch_size.json profiles/r06_pmc_fetch_size.json
# memory-side traffic of the HNSW search at the bench's own size (bench.py prints it while the sources' hash matches)
bash scripts/pmc_hnsw_traffic.sh 10000000 8192 128 > gpurun_out/r06_pmc_hnsw_traffic.stdout 2>&1; tail -5 gpurun_out/r06_pmc_hnsw_traffic.log
cp gpurun_out/r06_pmc_hnsw_traffic.json profiles/r06_pmc_hnsw_traffic.json
timeout 1500 python bench.py > gpurun_out/r06_bench_default_final.log 2> gpurun_out/r06_bench_default_final.err; tail -c 700 gpurun_out/r06_bench_default_final.log
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-serving --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $ROOT/gpurun_out/r06_bench_under_rocprof_final.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_final.csv
head -14 $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_final.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --dtype bf16 --no-cpu-baseline --no-serving --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $ROOT/gpurun_out/r06_bench_under_rocprof_bf16.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_bf16.csv
head -6 $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_bf16.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
timeout 1200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-serving --hybrid-rows 0 --bf16-rows 0 --single-query-steps 0 --steps 3 > $ROOT/gpurun_out/r06_bench_under_rocprof_hnsw.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_hnsw_10M.csv
head -8 $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_hnsw_10M.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
cd $ROOT
bash scripts/step_trace.sh r06_final_10M > /dev/null 2>&1; cat gpurun_out/r06_final_10M_step_trace.log | cut -c1-120
bash scripts/step_trace.sh r06_final_10M_bf16 --dtype bf16 > /dev/null 2>&1; cat gpurun_out/r06_final_10M_bf16_step_trace.log | cut -c1-120
bash scripts/step_trace.sh r06_final_1250k --rows 1250000 > /dev/null 2>&1; cat gpurun_out/r06_final_1250k_step_trace.log | cut -c1-120
python scripts/rerank_stamps.py 10000000 2>&1 | grep -v amdgpu.ids | tail -11 > gpurun_out/r06_rerank_stamps.log; cat gpurun_out/r06_rerank_stamps.log
python scripts/two_in_flight.py 10000000 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r06_two_in_flight.log; python scripts/two_in_flight.py 1250000 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r06_two_in_flight.log; cat gpurun_out/r06_two_in_flight.log
timeout 300 python scripts/fanout_probe.py --rows 10000000 --shards 8 > gpurun_out/r06_fanout_probe_threads.json 2>/dev/null; cat gpurun_out/r06_fanout_probe_threads.json
timeout 1500 python bench.py --gpus 8 --same-device --bf16-rows 1250000 --steps 20 --warmup 5 > gpurun_out/r06_bench_8_logical_shards.log 2> gpurun_out/r06_bench_8_logical_shards.err; tail -c 400 gpurun_out/r06_bench_8_logical_shards.log
