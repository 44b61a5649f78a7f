#!/bin/bash
# A/B of one environment switch on the same box: tools/r02_env_ab.sh VAR a b [a b ...]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 400 python bench.py --gaussians ${NG:-30000000} --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/ab_$v.log 2>&1
  grep -h '^{' $D/ab_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/ab_$v.log
done
