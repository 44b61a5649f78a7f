#!/bin/bash
# round 6, call 26: kernel trace of the multi-GPU step's exchange, one rank over RCCL, 30 M Gaussians
mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in sparse dense; do
rm -rf gpurun_out/xtrace_$mode
LOGRAST_DIST_SINGLE_RANK=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xtrace_$mode -o x -- python bench.py --steps 2 --warmup 1 --no-graphs --no-kernel-timing --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange $mode > gpurun_out/xtrace_$mode.log 2>&1
echo "$mode rc=$?"
f=$(find gpurun_out/xtrace_$mode -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-160
find gpurun_out/xtrace_$mode -name '*kernel_trace.csv' -delete
done
