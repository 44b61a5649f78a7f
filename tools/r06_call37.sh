#!/bin/bash
# round 6, call 37: the one-rank RCCL bench at 200 K repeated (the hint check on the direct sums)
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
LOGRAST_DIST_SINGLE_RANK=1 timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --gaussians 200000 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange sparse --streams 1 --full-out gpurun_out/rep.json > gpurun_out/rep.out 2> gpurun_out/rep.err
rc=$?
python -c "
import json; r=json.load(open('gpurun_out/rep.json')); print('run $i rc=$rc', r['exchange']['hint_check'])"
done
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2
