#!/bin/bash
# round 5, seventh GPU call: the chain rule over a compact live list (LOGRAST_PBWD_LIST): parity + A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropin_modes.py tests/test_gpu_knobs.py tests/test_gpu_graphs.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r05_tests_a.log 2>&1; echo "tests A rc=$?"; tail -6 gpurun_out/r05_tests_a.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -k "c5_band_full or image_split" > gpurun_out/r05_tests_b.log 2>&1; echo "tests B rc=$?"; tail -4 gpurun_out/r05_tests_b.log
rm -f gpurun_out/probe_list.jsonl
for l in 0 1; do
  timeout 300 python tools/kernel_probe.py --sink --views 2 --env LOGRAST_PBWD_LIST=$l --tag headline_list$l >> gpurun_out/probe_list.jsonl 2>> gpurun_out/probe_list.err
  timeout 300 python tools/kernel_probe.py --sink --views 2 --opacity -1 --env LOGRAST_PBWD_LIST=$l --tag rand_list$l >> gpurun_out/probe_list.jsonl 2>> gpurun_out/probe_list.err
  timeout 300 python tools/kernel_probe.py --sink --views 2 --scene trained --env LOGRAST_PBWD_LIST=$l --tag trained_list$l >> gpurun_out/probe_list.jsonl 2>> gpurun_out/probe_list.err
done
cat gpurun_out/probe_list.jsonl
LOGRAST_PBWD_LIST=0 timeout 300 python tools/bench_c5_band.py > gpurun_out/c5_list0.log 2>&1; tail -1 gpurun_out/c5_list0.log | cut -c1-1200
timeout 300 python tools/bench_c5_band.py > gpurun_out/c5_list1.log 2>&1; tail -1 gpurun_out/c5_list1.log | cut -c1-1200
