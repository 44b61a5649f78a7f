#!/bin/bash
# round 6, call 38: compiler-flag variants against the default build, same box (tools/kernel_probe.py --lib)
mkdir -p gpurun_out
: > gpurun_out/sched_ab.jsonl
for rep in 1 2; do
for v in "" _relax _track _nopost _nocluster; do
for op in 0.999 -1; do
timeout 300 python tools/kernel_probe.py --lib log_amd/lib/liblograst$v.so --sink --opacity $op 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"lib$v op $op\", /" >> gpurun_out/sched_ab.jsonl
done; done; done
python - <<'P'
import json
rows=[json.loads(l) for l in open("gpurun_out/sched_ab.jsonl") if l.startswith("{")]
keys=["project","fill_keys","sort_keys","blend_fwd","blend_bwd","project_bwd"]
for r in rows:
    print("%-28s" % r.get("variant"), {x: round(r[x],1) for x in r if isinstance(r.get(x),(int,float)) and any(x.startswith(p) for p in keys)})
P
