#!/bin/bash
# round 5, second GPU call: parity of the wave-cooperative 5..16-tile rects + chunk mask of lr_count_huge_kernel, the tests
# call 1 did not reach, A/B of the new binning on the trained-like scene, bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export LOG_REFERENCE=$PWD/.reference_mount
timeout 900 python -m pytest tests/test_gpu_log_plumbing.py -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo "plumbing rc=$?"; grep -h "passed\|failed" gpurun_out/log_plumbing_gpu.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_robustness.py tests/test_gpu_dist.py tests/test_gpu_dropin_modes.py -q -m gpu > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -4 gpurun_out/r05_tests_a.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "c5_band_full or trained_like or tree_ordered or band_projection" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
for mc in 0 1; do
  timeout 300 python tools/kernel_probe.py --scene trained --sink --views 2 --env LOGRAST_MID_COOP=$mc --tag trained_midcoop$mc >> gpurun_out/probe_midcoop.jsonl 2>> gpurun_out/probe_midcoop.err
done
cat gpurun_out/probe_midcoop.jsonl
timeout 900 python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/b_default.log') if l.startswith('{')][-1])
print("headline", d["ms_per_view"], {k: round(v["avg_us"]) for k,v in d["kernels"].items()})
for m in ("pipelined_opacity_rand","pipelined_trained_like"):
    x=d["modes"][m]; print(m, x.get("ms_per_view"), {k: round(v["avg_us"]) for k,v in x.get("kernels",{}).items()})
s=d["secondary"]; print("c2", s["c2"]["modes"]["pipelined"]["ms_per_view"], "c3", s["c3"]["ms_per_view"], s["c3"].get("stages_ms"), "c5", s["c5_band"]["ms_per_view_band_clipped_gradient_sink"])
P
