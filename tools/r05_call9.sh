#!/bin/bash
# chain rule back at three / four waves per SIMD (no __restrict__ on the inlined row function): its tests + the bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dropin_modes.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r05_tests_c.log 2>&1; echo "tests C rc=$?"; tail -4 gpurun_out/r05_tests_c.log
timeout 900 python bench.py > gpurun_out/b_default_c9.log 2>&1; echo "bench rc=$?"
grep -h '^{' gpurun_out/b_default_c9.log | cut -c1-200
