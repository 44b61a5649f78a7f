#!/bin/bash
# lazily ordered lists at full size: the scale tests (30 M, 4K grid, 100 M band) + the parity cases again (second-pass grids)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lazily or ordered or forward_bit_exact" > gpurun_out/r05_tests_e.log 2>&1; echo "tests E rc=$?"; tail -4 gpurun_out/r05_tests_e.log
timeout 1200 python -m pytest tests/test_gpu_scale.py -q -m gpu > gpurun_out/r05_tests_f.log 2>&1; echo "tests F rc=$?"; tail -12 gpurun_out/r05_tests_f.log
timeout 300 python tools/kernel_probe.py --sink --views 2 > gpurun_out/probe_c11.log 2>&1; tail -1 gpurun_out/probe_c11.log
