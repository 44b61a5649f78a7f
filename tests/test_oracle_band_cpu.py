"""oracle.forward(tile_rows=...) -- the whole-view render restricted to a band of tile rows (what the C5 parity test on
the GPU compares lr_project_band_kernel with) -- against the oracle's own whole-view render: inside the band identical
pixels and maps; the kept Gaussians are exactly those whose rect reaches the band, with the whole view's records."""
import numpy as np

from util import cam_tan


def test_band_restricted_oracle_equals_the_whole_view_inside_the_band(oracle_mod):
    from log_amd import scenes
    W, H = 320, 208                                               # 13 tile rows
    cam = scenes.orbit_cameras(4, W=W, H=H, focal=300.0)[1]
    sc = scenes.random_scene(20000, seed=4, opacity=None, smax=0.05)
    tfx, tfy = cam_tan(cam)
    v = oracle_mod.make_view(W, H, tfx, tfy, cam["world_view_transform"], cam["full_proj_transform"], [0.1, 0.2, 0.3])
    args = (sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    full = oracle_mod.forward(v, *args)
    dL = np.random.default_rng(0).random((3, H, W), dtype=np.float32)
    covered = np.zeros(len(full["radii"]), bool)
    g_sum = None
    for b, e in ((0, 4), (4, 9), (9, 13)):
        band = oracle_mod.forward(v, *args, tile_rows=(b, e))
        py0, py1 = 16 * b, min(16 * e, H)
        for k in ("image",):
            assert np.array_equal(band[k][:, py0:py1].view(np.uint32), full[k][:, py0:py1].view(np.uint32))
        for k in ("final_T", "point_weight_pixel"):
            assert np.array_equal(band[k][py0:py1].view(np.uint32), full[k][py0:py1].view(np.uint32)), k
        assert np.array_equal(band["point_id_pixel"][py0:py1], full["point_id_pixel"][py0:py1])
        out = np.ones(H, bool)
        out[py0:py1] = False
        assert (band["image"][:, out] == np.array([0.1, 0.2, 0.3], np.float32)[:, None, None]).all()
        assert (band["n_contrib"][out] == 0).all() and (band["point_id_pixel"][out] == -1).all()
        r0, r1 = full["rec"][:, 10].view(np.uint32), full["rec"][:, 11].view(np.uint32)
        reach = (full["radii"] > 0) & ((r0 >> 16) < e) & ((r1 >> 16) > b)
        assert np.array_equal(band["radii"] > 0, reach)
        assert np.array_equal(band["radii"][reach], full["radii"][reach])
        assert np.array_equal(band["rec"][reach][:, :10].view(np.uint32), full["rec"][reach][:, :10].view(np.uint32))
        br0, br1 = band["rec"][reach][:, 10].view(np.uint32), band["rec"][reach][:, 11].view(np.uint32)
        assert ((br0 >> 16) >= b).all() and ((br1 >> 16) <= e).all() and ((br1 >> 16) > (br0 >> 16)).all()
        assert band["I"] == int(band["tiles_touched"].astype(np.int64).sum()) == int(band["tile_offsets"][-1])
        covered |= reach
        d = np.zeros_like(dL)
        d[:, py0:py1] = dL[:, py0:py1]
        g = oracle_mod.backward(v, band, d)
        g_sum = g if g_sum is None else {k: g_sum[k] + g[k] for k in g}
    assert np.array_equal(covered, full["radii"] > 0)
    g_full = oracle_mod.backward(v, full, dL)
    for k in g_full:                                              # the bands' gradients sum to the whole view's
        ref = np.linalg.norm(g_full[k])
        assert np.linalg.norm(g_sum[k] - g_full[k]) <= 2e-5 * ref + 1e-12, k
