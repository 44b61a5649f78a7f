"""Turns the rocprofv3 / bench outputs that a gpurun call left in gpurun_out/ into the committed summaries under
profiles/ (round tag as argument, default r01).  python tools/collect_profiles.py [r01]"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def stats(path, title, out):
    rows = list(csv.DictReader(open(path)))
    o = ["# " + title, "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        o.append(f"| `{name[:60]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                 f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | "
                 f"{float(r['Percentage']):.2f} |")
    o += ["", "(The forward compositing kernel is launched TWICE per view since round 5's lazily ordered lists: the first "
          "launch does the work, the second -- with `lr_sort_long_kernel<true>` in front of it: the tails of lists whose walk "
          "ran out of ordered entries -- finds nothing to do in this workload (min us column); its per-view time is avg x 2, "
          "which is what `bench.py`'s HIP events report as blend_fwd + lazy_tail.)"]
    open(out, "w").write("\n".join(o) + "\n")


for n, t in (("b_default", "default"), ("b_oprand", "opacity_rand"), ("b_10M", "10M")):
    f = os.path.join(G, n + ".log")
    if os.path.exists(f) and any(l.startswith("{") for l in open(f)):     # the compact contract line (what the driver parses)
        line = [l for l in open(f) if l.startswith("{")][-1]
        json.dump(json.loads(line), open(os.path.join(P, f"{tag}_bench_{t}_line.json"), "w"), indent=1)
    f = os.path.join(G, n + "_full.json")
    if os.path.exists(f):                                                 # the full result (bench.py --full-out)
        json.dump(json.load(open(f)), open(os.path.join(P, f"{tag}_bench_{'full' if t == 'default' else t}.json"), "w"), indent=1)
cmd = ("python bench.py --views 4 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-kernel-timing "
       "--no-secondary --no-dropin-mode --no-rand-variant --no-forward-only --no-trained-like")
WORKLOAD = "the bench headline: 30 M random Gaussians @1080p, 4 views (stats pass + 1 step), one stream, eager launches"
src = os.path.join(G, f"{tag}_trace", "h30_kernel_stats.csv")
if os.path.exists(src):
    stats(src, f"rocprofv3 --kernel-trace --stats -- {cmd}; {WORKLOAD}", os.path.join(P, f"{tag}_kernel_stats.md"))
    shutil.copy(src, os.path.join(P, f"{tag}_kernel_stats.csv"))
src_t = os.path.join(G, f"{tag}_trace_trained", "t30_kernel_stats.csv")
if os.path.exists(src_t):
    stats(src_t, f"rocprofv3 --kernel-trace --stats -- {cmd} --scene trained; the same step on the trained-like scene "
                 "(log_amd.scenes.trained_like_scene, 30 M Gaussians @1080p)", os.path.join(P, f"{tag}_kernel_stats_trained.md"))
pm = os.path.join(ROOT, "tools", "pmc_summary.py")
sq = os.path.join(G, f"{tag}_pmc_sq", "h30_counter_collection.csv")
fe = os.path.join(G, f"{tag}_pmc_fetch", "h30_counter_collection.csv")
wr = os.path.join(G, f"{tag}_pmc_write", "h30_counter_collection.csv")
if all(os.path.exists(x) for x in (sq, fe, wr)):
    out = ["# rocprofv3 --pmc (separate passes), mean per launch; " + WORKLOAD, "",
           "(The forward compositing kernel runs twice per view -- the second launch, the tails of lazily ordered lists, is "
           "idle here: its per-launch means below are HALF its per-view sums.)", "", "## SQ",
           subprocess.check_output([sys.executable, pm, sq], text=True), "",
           "## TCC FETCH_SIZE / WRITE_SIZE (KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, see "
           "MI355X_MICROARCH.md)", subprocess.check_output([sys.executable, pm, fe, wr], text=True)]
    for name, title in (("mix", "instruction mix (SQ_INSTS_VALU_*_F32: wave instructions)"),
                        ("mix2", "instruction mix (integer / scalar memory / LDS / vector stores)"), ("tcc", "L2 requests (TCC_*_sum)")):
        extra = os.path.join(G, f"{tag}_pmc_{name}", "h30_counter_collection.csv")
        if os.path.exists(extra):
            out += ["", "## " + title, subprocess.check_output([sys.executable, pm, extra], text=True)]
    open(os.path.join(P, f"{tag}_pmc.md"), "w").write("\n".join(out))

    def load(path, counter):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k] += float(r["Counter_Value"])
                cnt[k] += 1
        # per launch -- except the two kernels the lazily ordered lists launch twice per view (second launch idle): per view
        twice = lambda k: k.startswith("lr_blend_fwd")
        return {k: agg[k] / (cnt[k] / 2 if twice(k) and cnt[k] % 2 == 0 else cnt[k]) for k in agg}
    f, w = load(fe, "FETCH_SIZE"), load(wr, "WRITE_SIZE")
    names = {"lr_blend_bwd_kernel": "blend_bwd", "lr_blend_bwd_rows_kernel": "blend_bwd", "lr_blend_bwd_kernel<false>": "blend_bwd",
             "lr_blend_bwd_rows_kernel<false>": "blend_bwd", "lr_blend_bwd_kernel<true>": "blend_bwd",
             "lr_blend_bwd_rows_kernel<true>": "blend_bwd",      # (<true>: the reverse walk on the forward's hit masks -- the steady state; listed last: it wins)
             "lr_blend_fwd_kernel<true>": "blend_fwd",
             "lr_blend_fwd_rows_kernel<true>": "blend_fwd", "lr_project_kernel": "project", "lr_project_batched_kernel": "project",
             "lr_project_batched_kernel<false>": "project", "lr_fill_staged_kernel": "fill_keys",
             "lr_fill_kernel": "fill_keys", "lr_fill_kernel<1>": "fill_keys", "lr_project_bwd_kernel<true, true, false, true>": "project_bwd",
             "lr_project_bwd_kernel<true, true, false, true, false>": "project_bwd",
             "lr_project_bwd_kernel<true, true, false, true, true>": "project_bwd",
             "lr_project_bwd_kernel<true, true, false, true, true, 1>": "project_bwd",
             "lr_fill_staged_kernel<2>": "fill_keys", "lr_fill_staged_kernel<1>": "fill_keys", "lr_fill_staged_kernel<3>": "fill_keys",
             "lr_sort_long_kernel": "sort", "lr_sort_long_kernel<false>": "sort"}
    tj = os.path.join(P, f"{tag}_traffic_30M.json")
    d = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes): " + cmd,
         "correction": "traffic_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE "
                       "tallies the 128-byte requests of wide coalesced reads at 64 B; the factor is exact for the streaming "
                       "kernels and an upper bound for the gathers of the compositing kernels)",
         "note": "30 M Gaussians: 1.7 GB of inputs + 1.9 GB of records per view, far past the 256 MiB Infinity Cache, so the "
                 "counters are memory-side traffic (SURVEY 8d)",
         "workload": {"gaussians": 30000000, "width": 1920, "height": 1080}}
    d["kernels"] = {s: {"fetch_kb": f[k], "write_kb": w[k], "traffic_bytes": (2 * f[k] + w[k]) * 1024}
                    for k, s in names.items() if k in f and k in w}
    # VALU issue utilisation from the SQ pass: SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD; 1024 SIMDs, 2.4 GHz peak
    act, durs = load(sq, "SQ_ACTIVE_INST_VALU"), collections.defaultdict(list)
    for r in csv.DictReader(open(sq)):
        durs[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for k, sname in names.items():
        if k in act and sname in d["kernels"] and durs.get(k):
            # (time and counter over the SAME set of launches: load() reports per view for the twice-launched compositing kernel)
            rows = len({(r["Dispatch_Id"]) for r in csv.DictReader(open(sq)) if r["Kernel_Name"].split("(")[0].replace("void ", "") == k})
            per = rows // 2 if (k.startswith("lr_blend_fwd") and rows % 2 == 0) else rows
            avg = (sum(durs[k]) / len(durs[k])) * rows / per
            d["kernels"][sname]["valu_active_frac_at_2p4GHz"] = act[k] * 4.0 / (1024 * avg * 2.4e9)
    json.dump(d, open(tj, "w"), indent=1)
print(sorted(os.listdir(P)))
