"""The multi-process path of bench.py on ONE GPU: two ranks sharing the device over gloo (RCCL refuses two ranks on one
device; the driver's runs use RCCL with one GPU per rank).  Exercises the self-spawn of `bench.py --gpus N`, the
per-rank view sharding, the gradient bucket reduction and the n_gpus the line reports."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None):
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=e, capture_output=True,
                         text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[-1])


def test_bench_self_spawns_two_ranks_and_reports_them():
    one = _bench("--gpus", "1", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                 "--no-cpu-baseline", "--no-dropin-mode")
    two = _bench("--gpus", "2", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                 "--no-cpu-baseline", "--no-dropin-mode", env={"LOGRAST_DIST_BACKEND": "gloo", "LOGRAST_SHARE_GPU": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["views_per_gpu"] == one["config"]["views_per_gpu"] == 8       # weak scaling: per-GPU work is fixed
    assert "dp2" in two["config"]["parallelism"] and two["value"] > 0
    # both ranks rendered the same Gaussians from their own 8 of the 16 cameras
    assert abs(two["config"]["visible_per_view"] - one["config"]["visible_per_view"]) < 0.05 * one["config"]["visible_per_view"]


def test_bench_line_carries_the_contract_fields():
    d = _bench("--gaussians", "300000", "--steps", "2", "--warmup", "1", "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "modes"):
        assert k in d, k
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert set(d["modes"]) == {"pipelined", "dropin_default"}
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
