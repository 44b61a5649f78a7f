#!/bin/bash
# round-3 run H: new tests (walk_form hint, band sparse projection) + the bench with its C5 band leg
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03h
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_dropin_modes.py tests/test_gpu_dist.py \
  "tests/test_gpu_scale.py::test_image_split_into_tile_row_bands" \
  "tests/test_gpu_scale.py::test_band_prepass_selects_exactly_the_gaussians_the_band_keeps" \
  tests/test_gpu_graphs.py -q -m gpu > $D/pytest.log 2>&1
tail -5 $D/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $D/bench_default.json 2> $D/bench_default.err
tail -c 600 $D/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/view", d["ms_per_view"])
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
s = d.get("secondary", {})
for k, v in s.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if isinstance(b, (int, float))})
print(json.dumps(s.get("c5_band", {}), indent=0)[:2500])
PY
