#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03m
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_knobs.py -q -m gpu -x > $D/pytest.log 2>&1
tail -3 $D/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python tools/band_project_probe.py > $D/$name.json 2> $D/$name.err; tail -1 $D/$name.json | cut -c1-420; }
run band_k1 A=1
run band_k2 LOGRAST_FILL_PER_THREAD=2
run band_k4 LOGRAST_FILL_PER_THREAD=4
P="--no-cpu-baseline --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only"
for k in 1 2 4; do
  LOGRAST_FILL_PER_THREAD=$k timeout 600 python bench.py $P > $D/b30_fill$k.json 2> $D/b30_fill$k.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03m/b30_fill$k.json").read().strip().splitlines()[-1])
print("fill per thread $k:", round(d["ms_per_view"], 3), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
done
