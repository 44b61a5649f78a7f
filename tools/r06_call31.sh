#!/bin/bash
# round 6, call 31: the hinted exchange's self-check (200 K through the test, 30 M through bench)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "rccl" 2>&1 | tail -4
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange sparse --full-out gpurun_out/rr_hint_full.json > gpurun_out/rr_hint.out 2> gpurun_out/rr_hint.err
echo rc=$?
python - <<P
import json
r=json.load(open("gpurun_out/rr_hint_full.json"))
print(round(r["value"]/1e9,3), "G/s", round(r["ms_per_step"],2), "ms/step", json.dumps(r["exchange"]["hint_check"]))
P
