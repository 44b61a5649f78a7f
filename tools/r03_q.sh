#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03q
mkdir -p "$D"
timeout 600 python -m pytest "tests/test_gpu_knobs.py::test_every_knob_leaves_every_output_bit_identical" -q -m gpu -x > $D/pytest.log 2>&1; tail -3 $D/pytest.log
for c in 0 1; do
  LOGRAST_CURSOR_DENSE=$c timeout 120 python tools/fill_probe.py > $D/fill_dense$c.json 2> $D/fill_dense$c.err
  tail -1 $D/fill_dense$c.json | cut -c1-200
done
LOGRAST_CURSOR_DENSE=1 LOGRAST_FILL_ABLATE=4 timeout 120 python tools/fill_probe.py 2>/dev/null | tail -1 | cut -c1-120
LOGRAST_CURSOR_DENSE=1 LOGRAST_FILL_ABLATE=8 timeout 120 python tools/fill_probe.py 2>/dev/null | tail -1 | cut -c1-120
