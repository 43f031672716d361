cd /tmp && export TMPDIR=/tmp
R=/root/repo
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=$R/gpurun_out/pmc_gpool; rm -rf $D
  timeout -s KILL 600 rocprofv3 --pmc $CNT --kernel-include-regex "hnsw_search_gpool_kernel|flat_scan_kernel<1," -d $D --output-format csv -- python $R/scripts/hnsw_probe.py --rows 1250000 --nq 8192 --ef 128 --tombstone --calibrate > $D.log 2>&1
  echo "== $CNT"; grep -E "allow-set|calibration" $D.log
  python $R/scripts/pmc_agg.py $D | python -c "
import json,sys
j=json.load(sys.stdin)
for k,v in j.items():
    if 'gpool_kernel' in k or 'flat_scan_kernel<1,' in k: print(k[:70], {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
"
  rm -rf $D
done
