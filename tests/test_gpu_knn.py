"""N1: simple_knn._C.distCUDA2 drop-in vs an exact CPU oracle (scipy cKDTree, float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle(pts):
    from scipy.spatial import cKDTree
    p64 = pts.astype(np.float64)
    d, _ = cKDTree(p64).query(p64, k=4)          # column 0 is the point itself (distance 0)
    return (d[:, 1:] ** 2).mean(axis=1)


def _clouds():
    rng = np.random.default_rng(0)
    uni = rng.random((200_000, 3), dtype=np.float32) * np.array([4, 2, 1], np.float32)
    # COLMAP-like: dense surface patches + sparse far outliers that stretch the bounding box
    centers = rng.normal(size=(40, 3)) * 5
    clustered = np.concatenate([
        (centers[rng.integers(0, 40, 250_000)] + rng.normal(size=(250_000, 3)) * 0.05 *
         np.array([1, 1, 0.02])),
        rng.normal(size=(500, 3)) * 200]).astype(np.float32)
    dup = np.repeat(rng.random((5_000, 3), dtype=np.float32), 3, axis=0)   # exact duplicates -> zero distances
    small = rng.random((5, 3), dtype=np.float32)
    return {"uniform": uni, "clustered": clustered, "duplicates": dup, "five_points": small}


@pytest.mark.parametrize("name", ["uniform", "clustered", "duplicates", "five_points"])
def test_dist_cuda2_matches_exact_knn(name):
    from simple_knn._C import distCUDA2
    pts = _clouds()[name]
    out = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    ref = _oracle(pts)
    assert out.shape == ref.shape and np.isfinite(out).all()
    # fp32 distances vs float64: relative to the coordinate scale
    scale = float(np.abs(pts).max()) ** 2
    np.testing.assert_allclose(out, ref, rtol=2e-4, atol=2e-6 * scale)


def test_dist_cuda2_call_site_contract():
    """The two reference call sites do clamp_min(distCUDA2(xyz.cuda()), 1e-7) then sqrt (file.py:88-91)."""
    from simple_knn._C import distCUDA2
    xyz = torch.rand(10_000, 3)
    dist2 = torch.clamp_min(distCUDA2(xyz.cuda()), 1e-7)
    scales = torch.sqrt(dist2).cpu()
    assert scales.shape == (10_000,) and scales.min() > 0
    with pytest.raises(Exception, match="MI355X"):
        distCUDA2(xyz)
