#!/bin/bash
# round 6, call 29: the exchange's device kernels in isolation, the tests of the hinted pack, the one-rank RCCL step again
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "hint or pack" 2>&1 | tail -4
timeout 300 python tools/exchange_kernel_probe.py 2>&1 | tail -1 | tee gpurun_out/exchange_kernel_probe.json
for h in hint nohint; do
flag=""; [ $h = nohint ] && flag="--no-pack-hint"
LOGRAST_DIST_SINGLE_RANK=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange sparse $flag --full-out gpurun_out/rccl_one_rank_sparse_${h}_full.json > gpurun_out/rccl_one_rank_sparse_$h.out 2> gpurun_out/rccl_one_rank_sparse_$h.err
echo "$h rc=$?"
python - <<P
import json
r=json.load(open("gpurun_out/rccl_one_rank_sparse_${h}_full.json"))
print("$h", round(r["value"]/1e9,3), "G/s", round(r["ms_per_step"],2), "ms/step", json.dumps(r["exchange"]["timing_ms"]), r["exchange"]["exchange_only_ms_per_step"])
P
done
