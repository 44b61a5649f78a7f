#!/bin/bash
# one quick look: 30 M (and optionally 1 M) per-kernel times, single stream
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-q}
for n in 30000000 1000000; do
timeout 400 python bench.py --gaussians $n --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_${n}.log 2>&1
grep -h '^{' $D/${TAG}_${n}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$n', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/${TAG}_${n}.log
done
