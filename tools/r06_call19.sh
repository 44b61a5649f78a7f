#!/bin/bash
# round 6, call 19: both compositing forms with the hit masks, 30 M (opaque / rand / trained-like) and C2
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --sink --views 4 --reps 3"
for f in 1 0; do
  $P --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag "30M_opaque_rows$f"
  $P --opacity -1 --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag "30M_rand_rows$f"
  $P --scene trained --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag "30M_trained_rows$f"
  $P --gaussians 1000000 --views 8 --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag "C2_rows$f"
  $P --gaussians 1000000 --views 8 --opacity -1 --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag "C2_rand_rows$f"
done 2>/dev/null | tee gpurun_out/r06_forms.jsonl
