#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropin_modes.py tests/test_gpu_knobs.py tests/test_gpu_graphs.py tests/test_gpu_dist.py -q -m gpu > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -6 gpurun_out/r05_tests_a.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "c5_band_full or image_split or band_pre" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
