#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
for S in 2 3 4 6 8; do
  timeout 400 python bench.py --gaussians 1000000 --steps 20 --warmup 3 --streams $S --no-cpu-baseline --no-secondary --no-dropin-mode --no-kernel-timing > $D/stc2_$S.log 2>&1
  grep -h '^{' $D/stc2_$S.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2 streams $S', 'ms/view', round(d['ms_per_view'],4), 'graphs', d['modes']['pipelined'].get('hip_graphs'))" || tail -n 5 $D/stc2_$S.log
done
