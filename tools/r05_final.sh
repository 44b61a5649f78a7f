#!/bin/bash
# round 5: the whole GPU suite (reference tree staged: the plumbing tests run too) + everything profiles/ is built from
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/parity_stats
export LOG_REFERENCE=$PWD/.reference_mount
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -h "passed\|failed" gpurun_out/gpu_tests.log | tail -3
grep "gpu plumbing" gpurun_out/gpu_tests.log > gpurun_out/log_plumbing_gpu.log
unset LOG_REFERENCE
bash tools/profile_round.sh r05 2>&1 | tail -12
