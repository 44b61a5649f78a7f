#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python - <<'PY'
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_log_step as B
r = B.c3_pipeline(views=4, sh_degree=3)
print("default  ms/view %.3f" % r["ms_per_view"], {k: round(v, 3) for k, v in r["stages_ms"].items()})
if "ms_per_view_capacity_hint" in r:
    print("hinted   ms/view %.3f" % r["ms_per_view_capacity_hint"], {k: round(v, 3) for k, v in r["stages_ms_capacity_hint"].items()})
else:
    print("hinted: ", r.get("capacity_hint_overflowed"))
PY
