#!/bin/bash
# lazily ordered lists: parity tests, knob sweep, then the bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_dropin_modes.py tests/test_gpu_robustness.py -x -q -m gpu > gpurun_out/r05_tests_d.log 2>&1; echo "tests D rc=$?"; tail -25 gpurun_out/r05_tests_d.log
timeout 900 python bench.py > gpurun_out/b_default_c10.log 2>&1; echo "bench rc=$?"
grep -h '^{' gpurun_out/b_default_c10.log | cut -c1-200
