#!/bin/bash
# Round-3 evidence, one gpurun call: PMC traffic of the timed kernels (own pass), the default bench line, the rocprofv3
# kernel-stats summaries of the FLAT leg and of the 10M HNSW leg, TCC hit/miss passes of the candidate filter over f32 and
# bf16 rows, the fan-out probe and the 8-logical-shard bench line.  Everything lands in gpurun_out/ (copied to profiles/
# by hand afterwards).
set -x
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
bash scripts/pmc_traffic.sh 2>&1 | tail -12
cp gpurun_out/r03_pmc_fetch_size.json profiles/r03_pmc_fetch_size.json
timeout 1200 python bench.py > gpurun_out/r03_bench_default_final.log 2> gpurun_out/r03_bench_default_final.err; tail -c 700 gpurun_out/r03_bench_default_final.log
cd /tmp && export TMPDIR=/tmp
# kernel stats, FLAT leg (timed region + single-query scan only: the other legs launch the same kernels in other regimes)
rm -rf $ROOT/gpurun_out/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $ROOT/gpurun_out/r03_bench_under_rocprof_final.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_final.csv
head -14 $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_final.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
# kernel stats, bf16 rows (the configs[3] shard)
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --dtype bf16 --no-cpu-baseline --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $ROOT/gpurun_out/r03_bench_under_rocprof_bf16.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_bf16.csv
head -6 $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_bf16.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
# kernel stats, the 10M HNSW leg (configs[2])
timeout 1200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_final --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --hybrid-rows 0 --bf16-rows 0 --single-query-steps 0 --steps 3 > $ROOT/gpurun_out/r03_bench_under_rocprof_hnsw.log 2>&1
find $ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_hnsw_10M.csv
head -8 $ROOT/gpurun_out/r03_rocprofv3_kernel_stats_hnsw_10M.csv | cut -c1-150
rm -rf $ROOT/gpurun_out/prof_final
# TCC hits / misses of the candidate filter (own --pmc passes)
for DT in f32 bf16; do
  Dd=$ROOT/gpurun_out/pmc_tcc_$DT; rm -rf $Dd
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $Dd --output-format csv -- python $ROOT/bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 --hybrid-rows 0 --bf16-rows 0 > $Dd.log 2>&1
  python $ROOT/scripts/pmc_agg.py $Dd | python -c "
import json,sys
j=json.load(sys.stdin)
print(json.dumps({k:v for k,v in j.items() if 'flat_filter' in k}, indent=1))" > $ROOT/gpurun_out/r03_pmc_tcc_filter_$DT.json
  cat $ROOT/gpurun_out/r03_pmc_tcc_filter_$DT.json | head -30
  rm -rf $Dd
done
cd $ROOT
# the fan-out: 8 logical shards on this one GPU, enqueue threads on / off
timeout 300 python scripts/fanout_probe.py --rows 10000000 --shards 8 > gpurun_out/r03_fanout_probe_threads.json 2>/dev/null; cat gpurun_out/r03_fanout_probe_threads.json
VK_SHARD_THREADS=0 timeout 300 python scripts/fanout_probe.py --rows 10000000 --shards 8 --skip-unsharded > gpurun_out/r03_fanout_probe_serial.json 2>/dev/null; cat gpurun_out/r03_fanout_probe_serial.json
# the N > 1 bench line, 8 logical shards on one GPU (the driver's multi-GPU run uses 8 physical ones)
timeout 1500 python bench.py --gpus 8 --same-device --bf16-rows 1250000 --steps 20 --warmup 5 > gpurun_out/r03_bench_8_logical_shards.log 2> gpurun_out/r03_bench_8_logical_shards.err; tail -c 400 gpurun_out/r03_bench_8_logical_shards.log
# what bounds a stage of the candidate filter: the ablation matrix (f16 experiment kernels over f32 / bf16 rows, the bf16
# DMA kernel's own) and the A/B switches of the final-pass kernels, all on this one lease
timeout 600 python scripts/filter_ablate.py --steps 30 --ablate=-1,0,1,3,7,1031,12,15,115,119,127 2>/dev/null > gpurun_out/r03_filter_ablations.log; cat gpurun_out/r03_filter_ablations.log | cut -c1-120
VK_FILTER_ABLATE_DMA=1 timeout 300 python scripts/filter_ablate.py --dtypes bf16 --steps 30 --ablate=-1,0,1,3,8,11,115,120,123 2>/dev/null > gpurun_out/r03_filter_ablations_bf16_dma.log; cat gpurun_out/r03_filter_ablations_bf16_dma.log | cut -c1-120
( for e in "" "VK_FILTER_BDMA=0" "VK_FILTER_DMA=0" "VK_FILTER_DMA=0 VK_FILTER_BDMA=0" "VK_FILTER_BF16_MFMA=0" "VK_FILTER_BF16_MFMA=0 VK_FILTER_BDMA=0"; do
    echo "== switches: ${e:-(defaults)}"; env $e timeout 300 python scripts/filter_ablate.py --steps 30 --ablate=-1 2>/dev/null | cut -c1-120; done ) > gpurun_out/r03_filter_kernel_ab.log; cat gpurun_out/r03_filter_kernel_ab.log
timeout 300 python scripts/flat_l2_batch.py > gpurun_out/r03_flat_l2_batch.log 2>/dev/null; cat gpurun_out/r03_flat_l2_batch.log
