#!/bin/bash
# The stamped part of scripts/final_artifacts.sh alone (after a change that touches the stamped sources but no kernel): PMC traffic of
# the timed kernels (own passes), the default bench line, the rocprofv3 kernel-stats summary of the FLAT leg, smoke().
set -x
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
bash scripts/pmc_traffic.sh 2>&1 | tail -12
cp gpurun_out/r06_pmc_fetch_size.json profiles/r06_pmc_fetch_size.json
bash scripts/pmc_hnsw_traffic.sh 10000000 8192 128 > gpurun_out/r06_pmc_hnsw_traffic.stdout 2>&1; tail -5 gpurun_out/r06_pmc_hnsw_traffic.log
cp gpurun_out/r06_pmc_hnsw_traffic.json profiles/r06_pmc_hnsw_traffic.json
timeout 1500 python bench.py > gpurun_out/r06_bench_default_final.log 2> gpurun_out/r06_bench_default_final.err; tail -c 700 gpurun_out/r06_bench_default_final.log
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-serving --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $ROOT/gpurun_out/r06_bench_under_rocprof_final.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_final.csv
head -6 $ROOT/gpurun_out/r06_rocprofv3_kernel_stats_final.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
