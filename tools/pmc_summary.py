"""Summarises rocprofv3 --pmc counter_collection.csv files per kernel (mean per launch).
    python tools/pmc_summary.py gpurun_out/r01_pmc_sq/c2_counter_collection.csv [more.csv ...]"""
import collections
import csv
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "")
    return n[:48]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            dur[(k, r["Dispatch_Id"], p)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kd = collections.defaultdict(list)
    for (k, _, _), d in dur.items():
        kd[k].append(d)
    names = sorted({c for k in agg for c in agg[k]})
    print("| kernel | launches | avg us (under PMC) | " + " | ".join(names) + " |")
    print("|---|---:|---:|" + "---:|" * len(names))
    for k in sorted(agg, key=lambda k: -sum(kd[k])):
        if not (k.startswith("lr_") or "lr_" in k):
            continue
        n = max(cnt[k].values())
        row = [f"{agg[k][c] / cnt[k][c]:.4g}" if cnt[k][c] else "" for c in names]
        print(f"| `{k}` | {n} | {sum(kd[k]) / len(kd[k]):.1f} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
