#!/bin/bash
# 30 M per-kernel times for a list of library variants (LOGRAST_LIB)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
for v in "$@"; do
  lib=$PWD/log_amd/lib/liblograst${v:+_$v}.so
  [ "$v" = base ] && lib=$PWD/log_amd/lib/liblograst.so
  LOGRAST_LIB=$lib timeout 400 python bench.py --gaussians ${NG:-30000000} --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/var_$v.log 2>&1
  grep -h '^{' $D/var_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/var_$v.log
done
