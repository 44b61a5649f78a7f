#!/bin/bash
# quick iteration: GPU parity tests + 30 M / C2 single-stream per-kernel times
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-it}
(timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $D/${TAG}_pytest.log 2>&1; echo pytest_exit=$? >> $D/${TAG}_pytest.log)
tail -n 5 $D/${TAG}_pytest.log
timeout 400 python bench.py --gaussians 30000000 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_30M_s1.log 2>&1
timeout 400 python bench.py --gaussians 1000000 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_1M_s1.log 2>&1
timeout 400 python bench.py --gaussians 10000000 --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_10M_s1.log 2>&1
for f in 30M 10M 1M; do grep -h '^{' $D/${TAG}_${f}_s1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$f', 'ms/view', round(d['ms_per_view'],3), 'host', round(d['host_enqueue_ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/${TAG}_${f}_s1.log; done
