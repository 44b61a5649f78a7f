#!/bin/bash
# round 6, call 5: the bench's own step (graphs, 8 views) with and without hit masks on one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for m in 0 1 0 1; do
  LOGRAST_HIT_MASKS=$m python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode > gpurun_out/b_masks$m.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_masks$m.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("masks=$m", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"]), "fwd %.0f bwd %.0f pbwd %.0f" % (r["us_blend_fwd"], r["us_blend_bwd"], r["us_project_bwd"]))
P
done | tee gpurun_out/r06_masks_bench_ab.txt
