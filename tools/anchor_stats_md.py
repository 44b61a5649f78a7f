"""profiles/<tag>_gradient_anchor_stats.md from the per-test dumps of tests/gpu_util.py::gradient_anchor_stats
(gpurun_out/parity_stats/*.json, written by `pytest -m gpu` on the GPU box).
usage: python tools/anchor_stats_md.py r03 [source note]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
note = sys.argv[2] if len(sys.argv) > 2 else "the round's last full `pytest -m gpu` run"
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity_stats", "*.json")))
out = ["# End-to-end gradients against the float64 twin (MI355X, `pytest -m gpu`, default kernels)", "",
       "Source: `tests/gpu_util.py::gradient_anchor_stats`, dumped by every full-size / case test of `tests/test_gpu_scale.py` and",
       "`tests/test_gpu_parity.py` (%s; this table: `python tools/anchor_stats_md.py %s`).  Per chain-rule output: rows the" % (note, tag),
       "float64 twin leaves non-zero; share of them fp32 cannot know to 3e-5 (excluded from claim 1); rel-L2 over the others, HIP",
       "and fp32 oracle, against float64; rows violating `|hip - f64| <= 2 |oracle - f64| + 64 units`; the worst",
       "`(|hip - f64| - 2 |oracle - f64|)` in units; row errors in units (q50 / q99 / max), HIP and oracle.  A unit = the row's fp32",
       "error scale (amplified input round-off + fp32 evaluation error of the chain rule, `oracle.backward_f64`).", "",
       "ALL rows (round 5): rel-L2 over every row, nothing excluded -- HIP vs float64, fp32 oracle vs float64, HIP vs the fp32",
       "oracle; asserted <= 1e-4 on the realistic inputs (`tree_ordered_heavy_tailed`, `trained_like_*`), printed everywhere.", "",
       "| case | output | rows | excluded | rel-L2 well (HIP) | rel-L2 well (oracle) | ALL rows: HIP vs f64 | oracle vs f64 | HIP vs oracle | violations | worst excess | HIP err units q50/q99/max | oracle err units q50/q99/max |",
       "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|---|"]
q = lambda v: " / ".join("%.1f" % x if x < 100 else "%.0f" % x for x in v)
tot_rows = tot_viol = 0
worst = 0.0
for f in files:
    d = json.load(open(f))
    case = os.path.basename(f)[:-5]
    for k in ("means3D", "scales", "rotations"):
        if k not in d:
            continue
        s = d[k]
        out.append("| %s | %s | %d | %.2f %% | %.1e | %.1e | %.1e | %.1e | %s | %d | %.1f | %s | %s |" % (
            case, k, s["rows"], 100 * s["excluded_fraction"], s["rel_l2_well_hip"], s["rel_l2_well_oracle"],
            s["rel_l2_all_hip"], s["rel_l2_all_oracle"],
            ("%.1e" % s["rel_l2_all_hip_vs_oracle"]) if "rel_l2_all_hip_vs_oracle" in s else "-",
            s["row_bound_violations"], s["worst_row_excess_units"], q(s["hip_err_units_q50_q99_max"]),
            q(s["oracle_err_units_q50_q99_max"])))
        tot_rows += s["rows"]; tot_viol += s["row_bound_violations"]; worst = max(worst, s["worst_row_excess_units"])
    rw = " ".join("%s %.1e/%.1e" % (k, d[k]["rel_l2_hip"], d[k]["rel_l2_oracle"])
                  for k in ("means2D", "conic", "opacities", "colors") if k in d)
    out.append("| %s | reverse walk (HIP / oracle vs float64, rel-L2) | | | | | | | | | | %s | |" % (case, rw))
out += ["", "Totals: %d (case, output) rows checked, %d violations of the every-row bound, worst excess %.1f units."
        % (tot_rows, tot_viol, worst)]
path = os.path.join(ROOT, "profiles", "%s_gradient_anchor_stats.md" % tag)
open(path, "w").write("\n".join(out) + "\n")
print(path, len(files), "cases;", tot_rows, "rows;", tot_viol, "violations; worst excess", round(worst, 1))


# ---- a summary bench.py can carry (round-5 verdict, next #2d: the deviation from north_star's plain 1e-4 reported, not buried)
import re
groups = {}
for f in files:
    d = json.load(open(f))
    case = os.path.basename(f)[:-5]
    g = re.sub(r"_view\d+$", "", case)
    e = groups.setdefault(g, {"cases": 0})
    e["cases"] += 1
    for k in ("means3D", "scales", "rotations"):
        if k in d:
            for key, src in (("all_rows_hip_vs_oracle", "rel_l2_all_hip_vs_oracle"), ("all_rows_hip_vs_f64", "rel_l2_all_hip"),
                             ("all_rows_oracle_vs_f64", "rel_l2_all_oracle"), ("well_conditioned_rows_hip_vs_f64", "rel_l2_well_hip"),
                             ("excluded_fraction", "excluded_fraction")):
                if src in d[k]:
                    e.setdefault(k, {})[key] = max(e.get(k, {}).get(key, 0.0), d[k][src])
    for k in ("means2D", "conic", "opacities", "colors"):
        if k in d:
            e.setdefault(k, {})["hip_vs_oracle"] = max(e.get(k, {}).get("hip_vs_oracle", 0.0), d[k].get("rel_l2_hip_vs_oracle", 0.0))
            e[k]["hip_vs_f64"] = max(e[k].get("hip_vs_f64", 0.0), d[k]["rel_l2_hip"])
summary = {"source": "tests/gpu_util.py::gradient_anchor_stats on the MI355X (%s); per group of cases the MAXIMUM relative L2 over "
                     "its views; tolerance of BASELINE.json north_star: 1e-4" % note,
           "round": tag, "groups": groups}
spath = os.path.join(ROOT, "profiles", "%s_parity_summary.json" % tag)
json.dump(summary, open(spath, "w"), indent=1, sort_keys=True)
print(spath, len(groups), "groups")
