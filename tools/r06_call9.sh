#!/bin/bash
# round 6, call 9: the whole GPU suite
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/parity_stats
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -30 gpurun_out/gpu_tests.log
