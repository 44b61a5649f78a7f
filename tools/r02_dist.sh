#!/bin/bash
# the distributed tests on the device + a 2-rank (shared GPU, gloo) bench line + the N = 1 line, both small
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_graphs.py tests/test_gpu_train_ops.py -x -q --timeout 600 > $D/dist_pytest.log 2>&1
tail -n 15 $D/dist_pytest.log
LOGRAST_DIST_BACKEND=gloo LOGRAST_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --gaussians 2000000 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-dropin-mode > $D/dist_bench2.log 2>&1
grep -h '^{' $D/dist_bench2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 ranks', d['n_gpus'], d['ms_per_view'], d['config']['parallelism'], d['modes']['pipelined'].get('hip_graphs'))" || tail -n 20 $D/dist_bench2.log
timeout 600 python bench.py --gaussians 2000000 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-dropin-mode > $D/dist_bench1.log 2>&1
grep -h '^{' $D/dist_bench1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('1 rank', d['n_gpus'], d['ms_per_view'], d['modes']['pipelined'].get('hip_graphs'))" || tail -n 20 $D/dist_bench1.log
