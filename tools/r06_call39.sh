#!/bin/bash
# round 6, call 39: where the GPU suite's 13 minutes go
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/gpu_tests_durations.log 2>&1; echo rc=$?
grep -A45 "slowest" gpurun_out/gpu_tests_durations.log | cut -c1-150
tail -2 gpurun_out/gpu_tests_durations.log
