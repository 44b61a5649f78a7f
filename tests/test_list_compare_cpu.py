"""The list comparison the GPU parity tests use when the support cull is on (tests/gpu_util.py), exercised on CPU
with lists derived from the oracle: a correctly culled list passes, every kind of defect is counted."""
import numpy as np

import gpu_util
from util import oracle_view, small_case


def _culled_copy(of, keep):
    lens = np.diff(of["tile_offsets"].astype(np.int64))
    tile = np.repeat(np.arange(len(lens)), lens)
    new_len = np.bincount(tile[keep], minlength=len(lens))
    offs = np.concatenate([[0], np.cumsum(new_len)]).astype(np.uint32)
    # n_contrib counts positions in the shorter list: shift by the dropped entries in front of the last contributor
    H, W = of["n_contrib"].shape
    gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    t = (ys // 16) * gx + xs // 16
    dropped_before = np.concatenate([[0], np.cumsum(~keep)])
    o0 = of["tile_offsets"].astype(np.int64)[t]
    n = of["n_contrib"].astype(np.int64)
    n_new = n - (dropped_before[o0 + n] - dropped_before[o0])
    hf = dict(of)
    hf.update(tile_offsets=offs, point_list=of["point_list"][keep], I=int(keep.sum()), n_contrib=n_new.astype(np.int32))
    return hf


def test_culled_list_comparison(oracle_mod):
    cam, sc = small_case(n=400, W=96, H=80, focal=110.0, smax=0.12, opacity=None)
    v = oracle_view(oracle_mod, cam)
    of = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    of["_view"] = v
    support = oracle_mod.instance_support(v, of).astype(bool)
    assert 0 < (~support).sum() < len(support)          # the case has something to cull and something to keep
    keys = ("offsets_mismatch", "list_mismatch", "n_contrib_mismatch")
    good = gpu_util.compare_forward(_culled_copy(of, support), of)
    assert all(good[k] == 0 for k in keys), good
    assert good["culled_instances"] == int((~support).sum())
    # dropping only some of the unsupported entries is fine too (the cull is conservative)
    part = support.copy()
    part[np.flatnonzero(~support)[::2]] = True
    st = gpu_util.compare_forward(_culled_copy(of, part), of)
    assert all(st[k] == 0 for k in keys), st
    # a supported entry dropped -> counted
    bad = support.copy()
    bad[np.flatnonzero(support)[3]] = False
    assert gpu_util.compare_forward(_culled_copy(of, bad), of)["list_mismatch"] > 0
    # two entries of one tile swapped -> counted
    hf = _culled_copy(of, support)
    lens = np.diff(hf["tile_offsets"].astype(np.int64))
    t = int(np.argmax(lens))
    assert lens[t] >= 2
    pl = hf["point_list"].copy()
    a = int(hf["tile_offsets"][t])
    pl[a], pl[a + 1] = pl[a + 1], pl[a]
    hf["point_list"] = pl
    assert gpu_util.compare_forward(hf, of)["list_mismatch"] > 0
    # an entry that is not in the oracle's tile list -> counted
    hf = _culled_copy(of, support)
    pl = hf["point_list"].copy()
    pl[a] = np.uint32(sc["xyz"].shape[0] - 1) if pl[a] != sc["xyz"].shape[0] - 1 else np.uint32(0)
    hf["point_list"] = pl
    st = gpu_util.compare_forward(hf, of)
    assert st["list_mismatch"] > 0 or st["offsets_mismatch"] > 0
