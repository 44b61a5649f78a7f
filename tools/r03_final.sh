#!/bin/bash
# final state of the round: the whole GPU suite, smoke(), the default bench line
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03final
mkdir -p "$D"
timeout 2400 python -m pytest tests -q -m gpu > $D/pytest.log 2>&1
tail -5 $D/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -2 $D/smoke.log
timeout 1200 python bench.py > $D/bench_default.json 2> $D/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03final/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/view", d["ms_per_view"], "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
print({k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
s = d.get("secondary", {})
print("c2", s["c2"]["modes"]["pipelined"]["ms_per_view"], "c3", s["c3"]["ms_per_view"], "c5", s["c5_band"]["ms_per_view_band_clipped_gradient_sink"], s.get("error"))
print("modes", {k: round(v.get("ms_per_view"), 3) for k, v in d.get("modes", {}).items() if isinstance(v, dict)})
PY
