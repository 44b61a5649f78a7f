#!/bin/bash
# round 6, call 14: the forward's per-block support test: exact per 4x4 block (1) against quadrant-exact + bounding box per block (0)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for t in 1 0 1 0; do
  LOGRAST_FWD_BLOCK_TEST=$t python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode --full-out gpurun_out/b_fbt$t.json > gpurun_out/b_fbt$t.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_fbt$t.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("fwd_block_test=$t", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"]), "fwd %.0f bwd %.0f" % (r["us_blend_fwd"], r["us_blend_bwd"]))
P
done | tee gpurun_out/r06_fwd_block_test_ab.txt
