#!/bin/bash
# round 6, call 25: the RCCL branches on the device (one-rank process group) + the fixed fuzz slice + a single-rank bench at 30 M
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "rccl" 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5
for mode in dense sparse; do
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange $mode --full-out gpurun_out/rccl_one_rank_${mode}_full.json > gpurun_out/rccl_one_rank_$mode.log 2>&1
echo "$mode rc=$?"; tail -1 gpurun_out/rccl_one_rank_$mode.log | cut -c1-600
done
