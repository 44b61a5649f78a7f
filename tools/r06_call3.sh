#!/bin/bash
# round 6, call 3: hit masks A/B again (forward collects its masks in LDS), + the masked-walk tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --sink --views 4 --reps 3"
for m in 0 1 0 1; do $P --env LOGRAST_HIT_MASKS=$m --tag "30M_opaque_masks$m"; done 2>/dev/null | tee gpurun_out/r06_masks_ab2.jsonl
for m in 0 1; do $P --opacity -1 --env LOGRAST_HIT_MASKS=$m --tag "30M_rand_masks$m"; done 2>/dev/null | tee -a gpurun_out/r06_masks_ab2.jsonl
for m in 0 1; do $P --scene trained --env LOGRAST_HIT_MASKS=$m --tag "30M_trained_masks$m"; done 2>/dev/null | tee -a gpurun_out/r06_masks_ab2.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hit_masks or lazily or backward_vs_oracle or scale_modifier" > gpurun_out/r06_parity3.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r06_parity3.log
