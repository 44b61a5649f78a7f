#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03p
mkdir -p "$D"
timeout 900 python -m pytest "tests/test_gpu_scale.py::test_band_projection_at_scale" -q -m gpu > $D/pytest.log 2>&1
tail -15 $D/pytest.log
