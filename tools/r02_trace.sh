#!/bin/bash
# kernel-trace stats of one 30 M step (4 views, one stream, eager launches): per-kernel average durations
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-tr}
P="python bench.py --views 4 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-secondary --no-dropin-mode ${2:-}"
rm -rf $D/${TAG}_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace -o h30 -- $P > $D/${TAG}_trace.log 2>&1
f=$(ls $D/${TAG}_trace/*kernel_stats.csv $D/${TAG}_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-70s calls %4s avg_us %9.1f total_us %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
