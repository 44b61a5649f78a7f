#!/bin/bash
# round 5, fifth GPU call: hybrid cooperative / per-lane counting of the 5..16-tile rects: parity + the threshold on the
# tree-ordered C3 view and the trained-like scene
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export LOG_REFERENCE=$PWD/.reference_mount
timeout 900 python -m pytest tests/test_gpu_log_plumbing.py -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo "plumbing rc=$?"; grep -h "passed\|failed" gpurun_out/log_plumbing_gpu.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_robustness.py -q -m gpu > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -4 gpurun_out/r05_tests_a.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "trained_like_1M or tree_ordered or band_projection" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
rm -f gpurun_out/probe_coopmax.jsonl
for mc in 0 8 16 24 64; do
  LOGRAST_MID_COOP=$mc timeout 600 python tools/bench_log_step.py 40000 7 3 4 > gpurun_out/c3_step_$mc.log 2>&1
  python - "$mc" <<'P'
import json,sys
mc=sys.argv[1]
d=json.loads([l for l in open('gpurun_out/c3_step_%s.log'%mc) if l.startswith('{')][-1])
k=d["kernels_us_per_view"]; print("c3 MID_COOP=%s ms/view %.3f project %.1f fill %.1f count_huge %.1f" % (mc, d["ms_per_view"], k["project"], k["fill_keys"], k["count_huge"]))
P
  timeout 300 python tools/kernel_probe.py --scene trained --sink --views 2 --fwd-only --env LOGRAST_MID_COOP=$mc --tag trained_coop$mc >> gpurun_out/probe_coopmax.jsonl 2>> gpurun_out/probe_coopmax.err
done
cat gpurun_out/probe_coopmax.jsonl
