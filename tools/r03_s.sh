#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03s
mkdir -p "$D"
timeout 900 python -m pytest tests/test_gpu_dropin_modes.py tests/test_gpu_dist.py tests/test_gpu_graphs.py -q -m gpu > $D/pytest.log 2>&1
tail -4 $D/pytest.log
P="--no-cpu-baseline --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only"
run() { name=$1; shift; timeout 600 python bench.py $P "$@" > $D/$name.json 2> $D/$name.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03s/$name.json").read().strip().splitlines()[-1])
print("$name:", round(d["ms_per_view"], 3), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
}
run rows
run planar --planar-bucket
run rows2
run planar2 --planar-bucket
run rows_rand --opacity -1
run planar_rand --opacity -1 --planar-bucket
