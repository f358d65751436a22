"""GPU (-m gpu): simple_knn._C.distCUDA2 drop-in (reference scene/gaussian_model.py:21,140,517) against an
exact k-d tree on the CPU (scipy cKDTree in float64 on the same float32 points).  The search is exact, so
only the fp32 rounding of the squared distances separates the two."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _want(pts):
    from scipy.spatial import cKDTree
    p = pts.astype(np.float64)
    d, _ = cKDTree(p).query(p, k=4)
    return (d[:, 1:] ** 2).mean(1)


@pytest.mark.parametrize("n,kind", [(4, "uniform"), (5, "uniform"), (777, "uniform"), (100_000, "uniform"),
                                    (200_000, "street"), (50_000, "clustered"), (30_000, "plane")])
def test_distcuda2_matches_exact_knn(n, kind):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    if kind == "uniform":
        pts = rng.uniform(-3, 3, (n, 3))
    elif kind == "street":                     # long thin point cloud like a KITTI-360 segment
        pts = np.stack([rng.uniform(0, 400, n), rng.normal(0, 6, n), rng.normal(0, 2, n)], 1)
    elif kind == "clustered":                  # wildly different densities
        c = rng.uniform(-50, 50, (20, 3))
        pts = c[rng.integers(0, 20, n)] + rng.normal(0, 1, (n, 3)) * rng.choice([0.01, 0.3, 5.0], (n, 1))
    else:                                      # degenerate extent along z
        pts = np.concatenate([rng.uniform(-10, 10, (n, 2)), np.zeros((n, 1))], 1)
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    want = _want(pts)
    assert got.shape == (n,) and got.dtype == np.float32
    scale = np.maximum(want, (np.abs(pts).max() ** 2) * 1e-6)   # cancellation floor of fp32 differences
    assert np.max(np.abs(got - want) / scale) < 1e-3
    assert np.median(np.abs(got - want) / np.maximum(want, 1e-30)) < 1e-5


def test_distcuda2_with_duplicate_points_and_argument_checks():
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(0)
    base = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    pts = np.concatenate([base, base[:100], base[:50]])          # exact duplicates -> zero distances
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    want = _want(pts)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-9)
    assert (got[:50] < want[:50] + 1e-9).all()
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 3))
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 2, device="cuda:0"))
    assert distCUDA2(torch.zeros(0, 3, device="cuda:0")).shape == (0,)
