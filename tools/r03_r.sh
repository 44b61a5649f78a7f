#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03r
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_parity.py "tests/test_gpu_scale.py::test_band_projection_at_scale" -q -m gpu -x > $D/pytest.log 2>&1
tail -3 $D/pytest.log
P="--no-cpu-baseline --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only"
for k in 1 2; do
  timeout 600 python bench.py $P > $D/b30_$k.json 2> $D/b30_$k.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03r/b30_$k.json").read().strip().splitlines()[-1])
print("run $k:", round(d["ms_per_view"], 3), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
done
timeout 300 python tools/band_project_probe.py 2>/dev/null | tail -1 | cut -c1-300
