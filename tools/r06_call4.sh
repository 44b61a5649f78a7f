#!/bin/bash
# round 6, call 4: where do the forward's 50 us with hit masks go?  (experiment build: LOGRAST_FWD_ABLATE 8 = no mask stores, 24 = no LDS writes either)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --views 4 --reps 3 --train-fwd-only --lib log_amd/lib/liblograst_exp.so"
for v in "LOGRAST_HIT_MASKS=0 LOGRAST_FWD_ABLATE=0" "LOGRAST_HIT_MASKS=1 LOGRAST_FWD_ABLATE=0" "LOGRAST_HIT_MASKS=1 LOGRAST_FWD_ABLATE=8" "LOGRAST_HIT_MASKS=1 LOGRAST_FWD_ABLATE=24" "LOGRAST_HIT_MASKS=0 LOGRAST_FWD_ABLATE=0" "LOGRAST_HIT_MASKS=1 LOGRAST_FWD_ABLATE=0"; do
  $P --env $v --tag "$v"; done 2>/dev/null | tee gpurun_out/r06_masks_where.jsonl
