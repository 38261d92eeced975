"""GPU: distCUDA2 against the reference's SciPy twin semantics
(/root/reference/scene/gaussian_model_ht.py:31-36), restated with scipy.spatial.KDTree here."""
import numpy as np
import pytest
import torch
from scipy.spatial import KDTree

pytestmark = pytest.mark.gpu


def _ref(points):
    d, _ = KDTree(points).query(points, k=min(4, len(points)))
    return (d[:, 1:] ** 2).mean(1)


@pytest.mark.parametrize("n,kind", [(2, "uniform"), (5, "uniform"), (1000, "uniform"), (20000, "clustered"), (200000, "surface"),
                                    (3000, "duplicates")])
def test_dist_cuda2_matches_kdtree(n, kind):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    if kind == "uniform":
        pts = rng.uniform(-1, 1, (n, 3))
    elif kind == "clustered":
        c = rng.uniform(-5, 5, (20, 3))
        pts = c[rng.integers(0, 20, n)] + 0.05 * rng.standard_normal((n, 3))
    elif kind == "surface":
        u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(-1, 1, n)
        pts = np.stack([np.cos(u) * np.sqrt(1 - v * v), np.sin(u) * np.sqrt(1 - v * v), v], 1) * 3.0
    else:
        pts = rng.uniform(-1, 1, (n // 3, 3)).repeat(3, 0)
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    if n >= 4:
        ref = _ref(pts.astype(np.float64))
    else:
        d = ((pts[:, None].astype(np.float64) - pts[None].astype(np.float64)) ** 2).sum(-1)
        ref = np.array([np.sort(d[i])[1:].mean() for i in range(n)])
    assert got.shape == (n,)
    assert np.abs(got - ref).max() <= 1e-5 * max(1e-12, np.abs(ref).max()) + 1e-10


def test_dist_cuda2_rejects_cpu():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))
