"""GPU nearest-neighbour kernel (`hm_nn_distance`) against scipy's cKDTree -- the host
implementation of the same query the reference makes through Open3D (metrics_3d/chamfer_distance.py:16-26)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def clouds(n, m, seed):
    rs = np.random.RandomState(seed)
    u = rs.randn(n, 3); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rs.randn(m, 3); v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.array([0.01, -0.02, 0.5])                       # a fruit half a metre from the origin
    return c + 0.040 * u * (1 + 0.05 * rs.randn(n, 1)), c + 0.041 * v


@pytest.mark.parametrize("n,m", [(1, 1), (7, 2049), (1000, 3), (5000, 4097), (100000, 60000)])
def test_nn_distance_matches_kdtree(n, m):
    from scipy.spatial import cKDTree
    from hortimapping_amd import metrics as MX
    a, b = clouds(n, m, n + m)
    ref = cKDTree(b).query(a)[0]
    got = MX.nn_distance_gpu(a, b)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-8 + 1e-5 * ref.max()      # fp32 differences on a centred 8 cm cloud


def test_empty_clouds_and_metric_classes():
    from hortimapping_amd import metrics as MX
    a, b = clouds(3000, 2500, 5)
    assert MX.nn_distance_gpu(a[:0], b).shape == (0,)
    assert np.isinf(MX.nn_distance_gpu(a, b[:0])).all()
    cd_h, cd_g = MX.ChamferDistance(), MX.ChamferDistance(backend="gpu")
    pr_h, pr_g = MX.PrecisionRecall(0.001, 0.01, 100), MX.PrecisionRecall(0.001, 0.01, 100, backend="gpu")
    for m in (cd_h, cd_g, pr_h, pr_g):
        m.update(a, b)
    assert abs(cd_h.compute() - cd_g.compute()) < 1e-6 * cd_h.compute()
    h, g = pr_h.compute_at_threshold(0.005), pr_g.compute_at_threshold(0.005)
    assert all(abs(x - y) < 0.2 for x, y in zip(h[:3], g[:3])) and h[3] == g[3]    # percentages; ties at a threshold


def test_million_point_clouds_throughput():
    """The evaluation size of the reference (1,000,000 samples per mesh): finishes in well under a second per direction."""
    import torch
    from hortimapping_amd import metrics as MX
    a, b = clouds(1000000, 1000000, 9)
    MX.nn_distance_gpu(a[:1000], b[:1000])
    t = time.time()
    d = MX.nn_distance_gpu(a, b)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"1M x 1M nearest neighbours: {dt*1e3:.0f} ms incl. host packing")
    assert np.isfinite(d).all() and d.max() < 0.02 and dt < 5.0
