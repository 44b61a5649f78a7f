#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03t
mkdir -p "$D"
P="--no-cpu-baseline --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only"
for a in 0 1 2 3; do
  LOGRAST_FWD_ABLATE=$a timeout 600 python bench.py $P > $D/fa$a.json 2> $D/fa$a.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03t/fa$a.json").read().strip().splitlines()[-1])
print("fwd ablate $a:", round(d["ms_per_view"], 3), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items() if k in ("blend_fwd", "blend_bwd", "project_bwd")})
PY
done
