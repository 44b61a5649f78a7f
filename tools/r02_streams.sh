#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
for S in 1 2 3; do
  timeout 400 python bench.py --gaussians ${NG:-30000000} --steps 3 --warmup 1 --streams $S --no-cpu-baseline --no-secondary --no-dropin-mode --no-kernel-timing > $D/st_$S.log 2>&1
  grep -h '^{' $D/st_$S.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams $S', 'ms/view', round(d['ms_per_view'],3), 'graphs', d['modes']['pipelined'].get('hip_graphs'), 'frac', round(d.get('algorithmic_frac_of_measured_copy',0),3))" || tail -n 5 $D/st_$S.log
done
