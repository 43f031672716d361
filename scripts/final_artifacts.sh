set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_traffic.sh 2>&1 | tail -15
cp gpurun_out/r03_pmc_fetch_size.json profiles/r03_pmc_fetch_size.json
timeout 900 python bench.py > gpurun_out/r03_bench_default_final.log 2> gpurun_out/r03_bench_default_final.err; tail -c 600 gpurun_out/r03_bench_default_final.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_under_rocprof_final.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/r03_rocprofv3_kernel_stats_final.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03_rocprofv3_kernel_stats_final.csv | cut -c1-160
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_final
