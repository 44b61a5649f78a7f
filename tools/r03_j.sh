#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03j
mkdir -p "$D"
run() { name=$1; shift; env "$@" timeout 300 python tools/band_project_probe.py > $D/$name.json 2> $D/$name.err; tail -1 $D/$name.json | cut -c1-400; }
run base A=1
run abl1 LOGRAST_PROJECT_ABLATE=1
run abl3 LOGRAST_PROJECT_ABLATE=3
run b8k LOGRAST_BATCH=8192
run sorted PROBE_SORTED=1
run dense LOGRAST_BAND_SPARSE=0
