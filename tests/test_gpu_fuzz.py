"""A fixed slice of the randomised parity sweep (tools/fuzz_parity.py; the full 1200-case run of the round:
profiles/r06_fuzz_parity.md): scenes, cameras, flavours, paths, compositing forms and launch-shape knobs drawn per seed --
forward bit for bit the oracle's, gradients to the criteria of tests/gpu_util.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

SEEDS = [100000 + i for i in range(0, 24)] + [200000 + i for i in range(0, 12)] + [300000 + i for i in (239, 253, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250)]


@pytest.mark.parametrize("block", range(6))
def test_random_cases_vs_oracle(oracle_mod, block):
    import fuzz_parity as F
    ran = 0
    for seed in SEEDS[block * 8:(block + 1) * 8]:
        r = F.run_case(oracle_mod, seed)
        ran += "skipped" not in r
    assert ran >= 6


def test_random_cases_bands_and_bucket_accumulation():
    """The same draws through the packages' autograd drop-in: the image split into 2 / 3 / 5 / 8 bands of tile rows renders the
    whole image bit for bit (radii / point_weight: maxima over the bands; reverse-walk gradients: sums), and views
    accumulated by the backward kernels into a row-major / planar gradient bucket equal autograd's own accumulation."""
    import fuzz_parity as F
    banded = 0
    for seed in [400000 + i for i in range(0, 40)]:
        r = F.run_props(seed)
        banded += bool(r.get("bands"))
    assert banded >= 20
