#!/bin/bash
# round-3 run I: band projection with per-wave compaction -- tests + the C5 band leg alone
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03i
mkdir -p "$D"
timeout 600 python -m pytest tests/test_gpu_knobs.py \
  "tests/test_gpu_scale.py::test_image_split_into_tile_row_bands" \
  "tests/test_gpu_scale.py::test_band_prepass_selects_exactly_the_gaussians_the_band_keeps" \
  tests/test_gpu_dist.py -q -m gpu > $D/pytest.log 2>&1
tail -5 $D/pytest.log
timeout 600 python tools/bench_c5_band.py > $D/c5.json 2> $D/c5.err
tail -c 400 $D/c5.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03i/c5.json").read().strip().splitlines()[-1])
for k, v in d.items():
    if k not in ("workload", "note"):
        print(k, v)
PY
