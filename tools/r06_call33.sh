#!/bin/bash
# round 6, call 33: the default (--exchange auto) step through the one-rank RCCL group; the new fuzz test
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --full-out gpurun_out/rr_auto_full.json > gpurun_out/rr_auto.out 2> gpurun_out/rr_auto.err
echo rc=$?
python - <<P
import json
r=json.load(open("gpurun_out/rr_auto_full.json"))
print(round(r["value"]/1e9,3), "G/s", round(r["ms_per_step"],2), "ms/step", json.dumps(r["exchange"])[:1200])
P
