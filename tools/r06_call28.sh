#!/bin/bash
# round 6, call 28: kernel trace of the streamed sparse exchange with the pack hint, one rank over RCCL, 30 M Gaussians
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/xtrace_hint
LOGRAST_DIST_SINGLE_RANK=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xtrace_hint -o x -- python bench.py --steps 4 --warmup 1 --no-graphs --no-kernel-timing --no-cpu-baseline --no-dropin-mode --no-secondary --no-forward-only --no-rand-variant --no-trained-like --exchange sparse > gpurun_out/xtrace_hint.log 2>&1
echo "rc=$?"
python - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/xtrace_hint/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST step: find the last 8 lr_project_batched launches that are followed by exchange kernels; take the window from the
# 8th-last project launch of the timed loop.  Simpler: list the per-kernel totals over the window of the last 9 'lx_pack_rows' launches
idx = [i for i, r in enumerate(rows) if "lx_pack_rows_kernel<true" in r["Kernel_Name"]]
print("pack<true> launches:", len(idx))
# windows: between consecutive finishes (lx_unpack_rows_kernel<0>)
fin = [i for i, r in enumerate(rows) if "lx_unpack_rows_kernel<0>" in r["Kernel_Name"]]
print("gathers:", len(fin))
a, b = fin[-3], fin[-2]          # one full timed step (exchange-only legs come later)
tot = collections.OrderedDict()
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].replace("void ", "")[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t = tot.setdefault(n, [0, 0.0]); t[0] += 1; t[1] += d
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6
print("one step, wall %.2f ms, kernel sum %.2f ms" % (span, sum(v[1] for v in tot.values()) / 1e3))
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %4d %9.1f us  avg %8.1f" % (n, c, d, d / c))
P
find gpurun_out/xtrace_hint -name '*kernel_trace.csv' -delete
