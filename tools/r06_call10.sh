#!/bin/bash
# round 6, call 10: the streamed exchange on one GPU (two ranks over gloo) + the remaining failures of the full run
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_dropin_modes.py tests/test_gpu_knobs.py -q -m gpu -x > gpurun_out/r06_dist.log 2>&1; echo "dist rc=$?"; tail -8 gpurun_out/r06_dist.log
timeout 600 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "band_projection" >> gpurun_out/r06_dist.log 2>&1; echo "band rc=$?"; tail -3 gpurun_out/r06_dist.log
# the exposed part of the exchange, one group per view against one group per step: two ranks on this one GPU (gloo: link
# time is NOT what this measures -- pack / unpack / overlap mechanics are), 30 M Gaussians
for parts in 1 8; do
  LOGRAST_DIST_BACKEND=gloo LOGRAST_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-dropin-mode --exchange sparse --exchange-parts $parts --full-out gpurun_out/b_2ranks_parts$parts.json > gpurun_out/b_2ranks_parts$parts.log 2>&1
  python - <<P
import json
d=json.load(open("gpurun_out/b_2ranks_parts$parts.json"))
e=d["exchange"]
print("parts=$parts ms_per_step %.2f exchange_only %.2f" % (d["ms_per_step"], e["exchange_only_ms_per_step"]), e["timing_ms"], "bytes/rank/step", e["bytes_moved_per_rank_per_step"], "touched", e.get("touched_row_fraction"))
P
done 2>&1 | tee gpurun_out/r06_streamed_exchange_one_gpu.txt
