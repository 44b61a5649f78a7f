#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03t
mkdir -p "$D"
P="--no-cpu-baseline --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only"
for v in "" _r512 _r2048 _r4096; do
  LOGRAST_LIB=$PWD/log_amd/lib/liblograst$v.so timeout 600 python bench.py $P > $D/rows$v.json 2> $D/rows$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03t/rows$v.json").read().strip().splitlines()[-1])
print("variant '$v':", round(d["ms_per_view"], 3), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items() if k in ("blend_fwd", "blend_bwd", "project_bwd")})
PY
done
