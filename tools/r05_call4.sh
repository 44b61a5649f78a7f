#!/bin/bash
# round 5, fourth GPU call: ranked 5..16-tile rects (LOGRAST_MID_RANK): parity (lists bit for bit), A/B, C1 device test
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export LOG_REFERENCE=$PWD/.reference_mount
timeout 900 python -m pytest tests/test_gpu_log_plumbing.py -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo "plumbing rc=$?"; grep -h "passed\|failed" gpurun_out/log_plumbing_gpu.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_robustness.py tests/test_gpu_graphs.py tests/test_gpu_dropin_modes.py -q -m gpu > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -4 gpurun_out/r05_tests_a.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "trained_like or tree_ordered or band_projection or properties or c2_full" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
rm -f gpurun_out/probe_midrank.jsonl
for mr in 0 1; do
  timeout 300 python tools/kernel_probe.py --scene trained --sink --views 2 --env LOGRAST_MID_RANK=$mr --tag trained_midrank$mr >> gpurun_out/probe_midrank.jsonl 2>> gpurun_out/probe_midrank.err
done
timeout 300 python tools/kernel_probe.py --sink --views 2 --tag headline >> gpurun_out/probe_midrank.jsonl 2>> gpurun_out/probe_midrank.err
cat gpurun_out/probe_midrank.jsonl
timeout 600 python tools/bench_log_step.py > gpurun_out/c3_step.log 2>&1; tail -2 gpurun_out/c3_step.log | cut -c1-1500
