"""The bench line the driver parses (round-5 verdict, next #1: BENCH_r05.json.parsed was null because the single line had
grown to 26 KB).  bench.compact_line() builds the LAST stdout line from the full result; here it is built from committed
full results of earlier runs (profiles/r0*_bench_default.json: what bench.py printed then) and from a minimal one."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _full_results():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]_bench_default.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_full.json"))):
        with open(path) as f:
            text = f.read().strip()
        try:
            d = json.loads(text)                            # pretty-printed copy
        except ValueError:
            d = json.loads(text.splitlines()[-1])           # raw stdout of a run
        if "modes" in d:
            out.append((os.path.basename(path), d))
    return out


def _check(line, full):
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT, len(text)
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-6)
    assert back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert isinstance(back["config"]["workload"], str) and back["config"]["workload"]
    assert "model" not in back["config"]
    # the driver's record truncates keys at 40 characters and strings at ~128
    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                assert len(k) <= 40, k
                walk(v)
        elif isinstance(o, str):
            assert len(o) <= 120, o
    walk({k: v for k, v in back.items() if k != "mode"})
    assert len(back["mode"]) <= 160
    assert text.index('"mode"') < 120          # which mode `value` is, inside the first 200 characters
    return back


@pytest.mark.parametrize("name,full", _full_results())
def test_compact_line_from_committed_full_results(name, full):
    bench.flatten_for_the_driver(full)
    back = _check(bench.compact_line(full), full)
    for k in ROOFLINE:
        assert k in back["roofline"], k
    assert back["roofline"]["frac"] == pytest.approx(back["roofline"]["achieved"] / back["roofline"]["peak"], rel=1e-3)
    for k in CPU:
        assert k in back["cpu_baseline"], k
    assert back["config"]["ms_per_view"] > 0
    if back["roofline"]["traffic"] is not None and name >= "r06":     # (rounds 4-5 named no file: "profiles/r*_traffic*.json")
        assert back["roofline"]["traffic_source"].startswith("profiles/r") and back["roofline"]["traffic_source"].endswith(".json")
    if "dropin_default" in full["modes"]:
        assert back["config"]["dropin_default_ms_per_view"] >= back["config"]["ms_per_view"] * 0.9


def test_there_is_a_committed_full_result_to_test_with():
    assert _full_results(), "profiles/ holds no full bench result"


def test_compact_line_survives_a_result_without_optional_legs():
    full = {"metric": "Gaussians/sec fwd+bwd @1080p", "value": 1.0e9, "unit": "Gaussians/s", "n_gpus": 2, "steps": 3,
            "warmup": 1, "ms_per_step": 12.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "x" * 500, "gaussians": 1000, "width": 64, "height": 64, "views_per_gpu": 2,
                       "parallelism": "p" * 500},
            "modes": {"pipelined": {"ms_per_view": 6.25}},
            "exchange": {"mode": "sparse", "parts": 2, "exchange_only_ms_per_step": 1.25,
                         "bytes_moved_per_rank_per_step": 12345}}
    back = _check(bench.compact_line(full), full)
    assert "roofline" not in back and "cpu_baseline" not in back
    assert back["config"]["exchange_mode"] == "sparse" and back["config"]["exchange_parts"] == 2


def test_an_oversized_kernel_table_is_shed_not_printed():
    name, full = _full_results()[-1]
    bench.flatten_for_the_driver(full)
    for i in range(400):
        full["roofline"]["kernel_us_made_up_kernel_%03d" % i] = 1.0 + i
    line = bench.compact_line(full)
    assert len(json.dumps(line, separators=(",", ":"))) < bench.LINE_LIMIT
    assert "frac" in line["roofline"]


def test_emit_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    name, full = _full_results()[-1]
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.flatten_for_the_driver(full)
    bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    assert json.loads(out[-1])["value"] == pytest.approx(full["value"], rel=1e-6)
    assert len(out[-1]) < bench.LINE_LIMIT
    with open(tmp_path / "bench_full.json") as f:
        assert json.load(f)["value"] == full["value"]


def test_roofline_traffic_comes_from_this_round_or_not_at_all():
    """round-5 verdict, #8: `roofline.traffic` is read from a committed counter profile, not measured in the run -- so it
    must be this round's (the kernels of an earlier round's profile have changed) or absent."""
    path, d = bench.traffic_profile(30_000_000, 1920, 1080)
    assert os.path.exists(os.path.join(ROOT, "profiles", "r05_traffic_30M.json"))       # an older one exists, and is not taken
    assert path is None or os.path.basename(path).startswith("r%02d_" % bench.BENCH_ROUND), path
    if path is None:
        assert bench.pmc_traffic("project", 30_000_000, 1920, 1080) is None
    else:
        assert d["kernels"]["project"]["traffic_bytes"] > 3.0e9                         # >= the algorithmic 100 B per Gaussian
        assert bench.pmc_traffic("project", 30_000_000, 1920, 1080) == d["kernels"]["project"]["traffic_bytes"]
    assert bench.pmc_traffic("project", 12345, 64, 64) is None


def test_parity_summary_reaches_the_line():
    p = bench.parity_summary()
    if p is None:
        pytest.skip("no parity summary of this round committed yet")
    line = bench.parity_line(p)
    assert line["tol"] == 1e-4 and line["file"].startswith("profiles/r%02d_" % bench.BENCH_ROUND)
    # the upstream package's flavour meets north_star's plain tolerance on every C2 view; the fork's clamp does not on
    # check_gui's uniform draws -- both are REPORTED
    assert line["c2_upstream_pkg"] < 1e-4 < line["c2_wodilate_fork_clamp"]
    assert line["trained_like_30M"] < 1e-4


def test_the_one_rank_rccl_leg_reaches_the_line():
    """Round 6: the default N = 1 run also measures rank 0's N > 1 step through a one-rank RCCL group (a child process) and
    the line carries its step time and exchange form -- or, had the leg failed, its error, without touching the headline."""
    path = os.path.join(ROOT, "profiles", "r06_bench_full.json")
    full = json.load(open(path))
    leg = full.get("multi_gpu_step_one_rank_rccl")
    assert isinstance(leg, dict) and leg.get("backend") == "nccl" and leg["ms_per_step"] > full["ms_per_step"]
    assert leg["hint_check"]["rel_l2"] < 1e-3 and leg["hint_check"]["rows_differ"] <= 1e-3 * leg["hint_check"]["rows_scanning"]
    line = bench.compact_line(full)
    assert line["config"]["rccl_one_rank_step_ms"] == pytest.approx(leg["ms_per_step"], rel=1e-4)
    assert line["config"]["rccl_one_rank_exchange"] == "sparse x8"
    broken = dict(full, multi_gpu_step_one_rank_rccl={"error": "TimeoutExpired: x" * 40})
    line = bench.compact_line(broken)
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and len(line["config"]["rccl_one_rank_exchange"]) <= 100
    assert "rccl_one_rank_step_ms" not in line["config"]
