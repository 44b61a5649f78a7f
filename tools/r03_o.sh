#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03o
mkdir -p "$D"
timeout 600 python tools/adam_probe.py > $D/adam_final.json 2> $D/adam.err; tail -1 $D/adam_final.json
