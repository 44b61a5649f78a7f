#!/bin/bash
# round 6, call 18: the row-split reverse walk without its commits (experiment build, LOGRAST_BWD_ABLATE=1): what the memory-side
# atomics cost next to the VALU work (results are garbage, only the times count)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P="python tools/kernel_probe.py --sink --views 4 --reps 3 --lib log_amd/lib/liblograst_exp.so"
for ab in 0 1 0 1; do $P --env LOGRAST_BWD_ABLATE=$ab --tag "30M_opaque_bwd_ablate$ab"; done 2>/dev/null | tee gpurun_out/r06_bwd_atomics.jsonl
for ab in 0 1; do $P --opacity -1 --env LOGRAST_BWD_ABLATE=$ab --tag "30M_rand_bwd_ablate$ab"; done 2>/dev/null | tee -a gpurun_out/r06_bwd_atomics.jsonl
for ab in 0 1; do $P --gaussians 1000000 --views 8 --env LOGRAST_BWD_ROWS=1 LOGRAST_FWD_ROWS=1 LOGRAST_BWD_ABLATE=$ab --tag "C2_rows_bwd_ablate$ab"; done 2>/dev/null | tee -a gpurun_out/r06_bwd_atomics.jsonl
