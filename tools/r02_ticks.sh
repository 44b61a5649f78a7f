#!/bin/bash
# 30 M timing with the product library + one view with the phase-timing variant of the long sort (LOGRAST_LIB)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-tk}
timeout 400 python bench.py --gaussians 30000000 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_30M_s1.log 2>&1
grep -h '^{' $D/${TAG}_30M_s1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('30M', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/${TAG}_30M_s1.log
LOGRAST_LIB=$PWD/log_amd/lib/liblograst_ticks.so timeout 400 python bench.py --gaussians 30000000 --views 1 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_ticks.log 2>&1
grep -h "longsort" $D/${TAG}_ticks.log | sort | uniq -c | sort -rn | head -12
