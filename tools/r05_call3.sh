#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export LOG_REFERENCE=$PWD/.reference_mount
timeout 900 python -m pytest tests/test_gpu_log_plumbing.py -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo "plumbing rc=$?"; grep -h "passed\|failed" gpurun_out/log_plumbing_gpu.log | tail -3
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_dropin_modes.py -q -m gpu > gpurun_out/r05_tests_c.log 2>&1; echo "tests C rc=$?"; tail -4 gpurun_out/r05_tests_c.log
rm -f gpurun_out/probe_forms.jsonl
for f in 0 1; do
  timeout 300 python tools/kernel_probe.py --scene trained --sink --views 2 --env LOGRAST_FWD_ROWS=$f LOGRAST_BWD_ROWS=$f --tag trained_rows$f >> gpurun_out/probe_forms.jsonl 2>> gpurun_out/probe_forms.err
done
cat gpurun_out/probe_forms.jsonl
