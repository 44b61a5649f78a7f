#!/bin/bash
# round 6, call 38: LLVM scheduling strategies (max-ilp, max-memory-clause) against the default build, same box
mkdir -p gpurun_out
: > gpurun_out/sched_ab.jsonl
for rep in 1 2; do
for v in "" _ilp _clause; do
for op in 0.999 -1; do
timeout 300 python tools/kernel_probe.py --lib log_amd/lib/liblograst$v.so --sink --opacity $op --tag "lib${v}_op${op}" 2>/dev/null | tail -1 >> gpurun_out/sched_ab.jsonl
done; done; done
python - <<'P'
import json
rows=[json.loads(l) for l in open("gpurun_out/sched_ab.jsonl") if l.startswith("{")]
keys=["project","fill_keys","sort_keys","blend_fwd","blend_bwd","project_bwd"]
for r in rows:
    k={x:r.get(x) for x in r if isinstance(r.get(x),(int,float))}
    print(r.get("tag"), {x: round(k[x],1) for x in k if any(x.startswith(p) for p in keys)} )
P
