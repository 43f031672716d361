#!/bin/bash
# Engine clock and power while K4 runs (is the matrix peak quoted at 2.4 GHz the peak the kernel actually sees?)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
python bench.py --steps 300 --warmup 2 --no-cpu-baseline --single-query-steps 0 --hnsw-rows 0 > /tmp/b.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Current Socket" | sed 's/.*level: [0-9S]*: //; s/.*Power (W): /W=/' | tr '\n' ' '; echo
done | sort | uniq -c | sort -k2 | tail -40
wait $BP
tail -1 /tmp/b.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'])"
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power" | tr '\n' ' '; echo
