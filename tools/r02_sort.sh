#!/bin/bash
# sort experiments: parity of the list-producing paths + 30 M / layered timings
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-so}
timeout 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py -q --timeout 600 > $D/${TAG}_pytest.log 2>&1
tail -n 4 $D/${TAG}_pytest.log
timeout 600 python -m pytest tests/test_gpu_scale.py -q --timeout 600 -k "30M or tree or 10M" > $D/${TAG}_pytest2.log 2>&1
tail -n 3 $D/${TAG}_pytest2.log
timeout 400 python bench.py --gaussians 30000000 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_30M_s1.log 2>&1
timeout 400 python bench.py --gaussians 10000000 --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_10M_s1.log 2>&1
timeout 400 python bench.py --gaussians 1000000 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_1M_s1.log 2>&1
for f in 30M 10M 1M; do grep -h '^{' $D/${TAG}_${f}_s1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$f', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/${TAG}_${f}_s1.log; done
timeout 300 python tools/bench_layers.py > $D/${TAG}_layers.log 2>&1; tail -n 4 $D/${TAG}_layers.log
