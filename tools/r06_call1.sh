#!/bin/bash
# round 6, first GPU call: the hit-mask hand-over + flavour tests, then the bench (compact last line)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
nproc; free -g | head -2
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_graphs.py -q -m gpu -x > gpurun_out/r06_parity.log 2>&1; echo "parity rc=$?"; tail -12 gpurun_out/r06_parity.log
timeout 900 python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"; tail -3 gpurun_out/b_default.err
tail -1 gpurun_out/b_default.log | wc -c
tail -1 gpurun_out/b_default.log
