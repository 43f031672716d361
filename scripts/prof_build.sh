cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_build
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_build --output-format csv -- python $GRAFT_REPO_ROOT/scripts/build_probe.py --rows 1000000 --dim 768 --skip-host > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_build -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r02_rocprofv3_kernel_stats_hnsw_build_1M.csv
grep "vk::" $f | head -16 | awk -F'","' '{gsub(/"/,"",$1); printf "%-70s calls %6s total_ms %9.1f avg_us %8.1f\n", substr($1,1,70), $2, $3/1e6, $4/1000}'
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_build
