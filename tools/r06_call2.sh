#!/bin/bash
# round 6, call 2: hit masks A/B on one box (kernel times of one view, eager, one stream), rest of the parity file
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --sink --views 4 --reps 3"
for m in 0 1 0 1; do $P --env LOGRAST_HIT_MASKS=$m --tag "30M_opaque_masks$m"; done 2>/dev/null | tee gpurun_out/r06_masks_ab.jsonl
for m in 0 1; do $P --opacity -1 --env LOGRAST_HIT_MASKS=$m --tag "30M_rand_masks$m"; done 2>/dev/null | tee -a gpurun_out/r06_masks_ab.jsonl
for m in 0 1; do $P --scene trained --env LOGRAST_HIT_MASKS=$m --tag "30M_trained_masks$m"; done 2>/dev/null | tee -a gpurun_out/r06_masks_ab.jsonl
for m in 0 1 0 1; do $P --gaussians 1000000 --views 8 --reps 5 --env LOGRAST_HIT_MASKS=$m --tag "C2_masks$m"; done 2>/dev/null | tee -a gpurun_out/r06_masks_ab.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "empty or scratch or autograd or determin or capacity or fused or large_tile or cov3d or psnr" > gpurun_out/r06_parity2.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r06_parity2.log
