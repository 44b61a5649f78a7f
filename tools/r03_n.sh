#!/bin/bash
# round-3 run N: the whole GPU suite + the default bench line
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03n
mkdir -p "$D"
timeout 2400 python -m pytest tests -q -m gpu > $D/pytest.log 2>&1
tail -8 $D/pytest.log
timeout 1200 python bench.py > $D/bench_default.json 2> $D/bench_default.err
tail -c 300 $D/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03n/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/view", d["ms_per_view"])
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
print("roofline", d["roofline"])
print("cpu", d.get("cpu_baseline"))
s = d.get("secondary", {})
for k, v in s.items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if isinstance(b, (int, float))})
print("modes", {k: v.get("ms_per_view") for k, v in d.get("modes", {}).items() if isinstance(v, dict)})
PY
