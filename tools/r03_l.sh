#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03l
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_knobs.py \
  "tests/test_gpu_scale.py::test_image_split_into_tile_row_bands" \
  "tests/test_gpu_scale.py::test_band_prepass_selects_exactly_the_gaussians_the_band_keeps" \
  -q -m gpu -x > $D/pytest.log 2>&1
tail -3 $D/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python tools/band_project_probe.py > $D/$name.json 2> $D/$name.err; tail -1 $D/$name.json | cut -c1-420; }
run base A=1
run abl1 LOGRAST_PROJECT_ABLATE=1
run abl3 LOGRAST_PROJECT_ABLATE=3
run sorted PROBE_SORTED=1
run dense LOGRAST_BAND_SPARSE=0
