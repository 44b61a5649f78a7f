#!/bin/bash
# round 5, sixth GPU call: rank rows for one in four Gaussians, lr_count_huge_kernel's direct path: parity + probes
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_gpu_robustness.py tests/test_gpu_graphs.py -q -m gpu > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -4 gpurun_out/r05_tests_a.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "trained_like or tree_ordered or band_projection or c2_full or properties" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
rm -f gpurun_out/probe_final.jsonl
timeout 300 python tools/kernel_probe.py --scene trained --sink --views 2 --tag trained >> gpurun_out/probe_final.jsonl 2>> gpurun_out/probe_final.err
timeout 300 python tools/kernel_probe.py --gaussians 1000000 --sink --views 8 --reps 4 --tag c2 >> gpurun_out/probe_final.jsonl 2>> gpurun_out/probe_final.err
timeout 300 python tools/kernel_probe.py --gaussians 1000000 --sink --views 8 --reps 4 --env LOGRAST_MID_RANK=0 --tag c2_norank >> gpurun_out/probe_final.jsonl 2>> gpurun_out/probe_final.err
cat gpurun_out/probe_final.jsonl
timeout 600 python tools/bench_log_step.py 40000 7 3 4 > gpurun_out/c3_step.log 2>&1; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/c3_step.log') if l.startswith('{')][-1])
print("c3 ms/view %.3f" % d["ms_per_view"], d["kernels_us_per_view"])
P
