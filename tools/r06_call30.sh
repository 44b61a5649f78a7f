#!/bin/bash
# round 6, call 30: repeatability of the one-rank RCCL step, hinted / scanning pack, dense
mkdir -p gpurun_out
for h in nohint hint nohint hint dense; do
flag="--exchange sparse"; [ $h = nohint ] && flag="--exchange sparse --no-pack-hint"; [ $h = dense ] && flag="--exchange dense"
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like $flag --full-out gpurun_out/rr_${h}_full.json > gpurun_out/rr_$h.out 2> gpurun_out/rr_$h.err
python - <<P
import json
r=json.load(open("gpurun_out/rr_${h}_full.json"))
print("$h", round(r["value"]/1e9,3), "G/s", round(r["ms_per_step"],2), "ms/step", json.dumps(r["exchange"]["timing_ms"]), round(r["exchange"]["exchange_only_ms_per_step"],2))
P
done
