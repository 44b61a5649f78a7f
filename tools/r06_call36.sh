#!/bin/bash
# round 6, call 36: deferred visible counts (one pass per step): test, then the default step through the one-rank RCCL group
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "hint or rccl or one_process" 2>&1 | tail -3
for i in 1 2; do
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --full-out gpurun_out/rr_auto_full.json > gpurun_out/rr_auto.out 2> gpurun_out/rr_auto.err
echo rc=$?
python - <<P
import json
r=json.load(open("gpurun_out/rr_auto_full.json"))
e=r["exchange"]
print(round(r["value"]/1e9,3), "G/s", round(r["ms_per_step"],2), "ms/step", e["timing_ms"], e["hint_check"])
P
done
