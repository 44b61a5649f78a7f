#!/bin/bash
# round 5, first GPU call: the new parity tests + a baseline bench line (run ON the GPU box from the repo root)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
free -g | head -2; nproc
export LOG_REFERENCE=$PWD/.reference_mount
timeout 900 python -m pytest tests/test_gpu_log_plumbing.py -x -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo "plumbing rc=$?"; tail -5 gpurun_out/log_plumbing_gpu.log
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_dist.py tests/test_gpu_dropin_modes.py -q -m gpu -x \
  -k "c5_band_full or trained_like or tree_ordered or pack_and_unpack or contract or walk_form or rccl" > gpurun_out/r05_newtests.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r05_newtests.log
timeout 900 python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"
grep -h '^{' gpurun_out/b_default.log | cut -c1-600
