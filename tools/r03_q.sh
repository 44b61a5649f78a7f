#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03q
mkdir -p "$D"
for a in 0 4 8 12 2; do
  LOGRAST_FILL_ABLATE=$a timeout 120 python tools/fill_probe.py > $D/fill_abl$a.json 2> $D/fill_abl$a.err
  tail -1 $D/fill_abl$a.json | cut -c1-300; tail -2 $D/fill_abl$a.err | grep -v amdgpu.ids
done
