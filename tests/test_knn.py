"""simple-knn replacement: oracle pinned against scipy's exact KD-tree, HIP against the oracle."""
import numpy as np
import pytest
import torch

import oracle


def clouds():
    rng = np.random.default_rng(7)
    uni = rng.uniform(-1.3, 1.3, (5000, 3)).astype(np.float32)
    clustered = np.concatenate([rng.normal(c, 0.02, (700, 3)) for c in rng.uniform(-5, 5, (6, 3))]).astype(np.float32)
    dup = np.concatenate([uni[:50], uni[:50], uni[100:140]]).astype(np.float32)          # exact duplicates: distance 0 counts
    line = np.stack([np.linspace(0, 1, 300), np.zeros(300), np.zeros(300)], 1).astype(np.float32)   # degenerate axes
    return {"uniform": uni, "clustered": clustered, "duplicates": dup, "line": line}


def test_oracle_matches_exact_kdtree():
    from scipy.spatial import cKDTree
    for name, pts in clouds().items():
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        want = (d[:, 1:] ** 2).mean(1)
        got = oracle.knn_mean_dist2(pts)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-12, err_msg=name)


@pytest.mark.gpu
def test_hip_distcuda2_matches_oracle():
    from simple_knn._C import distCUDA2
    for name, pts in clouds().items():
        got = distCUDA2(torch.as_tensor(pts, device="cuda")).cpu().numpy()
        want = oracle.knn_mean_dist2(pts)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12, err_msg=name)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))                                  # CPU tensors are rejected, no fallback
    assert distCUDA2(torch.zeros(0, 3, device="cuda")).shape == (0,)


@pytest.mark.gpu
def test_hip_distcuda2_large_cloud_against_kdtree():
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    pts = np.random.default_rng(1).uniform(-1.3, 1.3, (300000, 3)).astype(np.float32)
    got = distCUDA2(torch.as_tensor(pts, device="cuda")).cpu().numpy()
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
    np.testing.assert_allclose(got, (d[:, 1:] ** 2).mean(1), rtol=2e-5)
