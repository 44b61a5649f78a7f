#!/bin/bash
# round 6, call 20: what the forward's per-visit memory operations cost (experiment build: LOGRAST_FWD_ABLATE 1 = no point_weight
# atomicMax, 2 = no accumulator-row clears, 3 = neither), row-split form, training forwards
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --views 4 --reps 3 --train-fwd-only --lib log_amd/lib/liblograst_exp.so"
for ab in 0 1 2 3 0 3; do $P --env LOGRAST_FWD_ABLATE=$ab --tag "30M_opaque_fwd_ablate$ab"; done 2>/dev/null | tee gpurun_out/r06_fwd_memops.jsonl
for ab in 0 3; do $P --opacity -1 --env LOGRAST_FWD_ABLATE=$ab --tag "30M_rand_fwd_ablate$ab"; done 2>/dev/null | tee -a gpurun_out/r06_fwd_memops.jsonl
