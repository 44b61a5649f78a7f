#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/exchange_local_probe.py 2>gpurun_out/r06_exchange_local.err | tee gpurun_out/r06_exchange_local.json; tail -3 gpurun_out/r06_exchange_local.err
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_dropin_modes.py tests/test_gpu_knobs.py -q -m gpu > gpurun_out/r06_dist.log 2>&1; echo "dist rc=$?"; tail -5 gpurun_out/r06_dist.log
