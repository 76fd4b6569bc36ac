"""CPU tests of the "next" rows around the hot path: voxel grid, data prep, metrics, PLY I/O."""
import numpy as np
import pytest
import torch

from tests.golden_util import load


def test_create_voxel_grid_matches_golden():
    """G11: create_voxel_grid (utils.py:542-562)."""
    from hortimapping_amd.mesher import create_voxel_grid
    g = load("g11_voxel_grid")
    assert np.allclose(create_voxel_grid(5).numpy(), g["n5"], atol=1e-7)
    assert np.allclose(create_voxel_grid(8).numpy(), g["n8"], atol=1e-7)


def test_get_render_data_matches_golden():
    """G10: get_render_data (utils.py:39-109) incl. the np.random.choice pixel sub-sampling under seed 42."""
    from hortimapping_amd.data_prep import get_render_data
    g = load("g10_data_prep")
    cfg = {"opt": {"render": {"n_fg_pix": int(g["n_fg_pix"]), "n_bg_pix": int(g["n_bg_pix"]), "n_bg_pad": int(g["n_bg_pad"])}}}
    idimg, depth = g["id_img"], g["depth_img"]
    np.random.seed(42)
    rd = get_render_data(5, {0: idimg, 3: idimg.T.copy()}, {0: depth, 3: depth.T.copy()}, {0: np.eye(4), 3: np.eye(4)},
                         (64, 64), np.linalg.inv(g["K64"]), cfg, min_pix_count_match=100, max_bbx_size=300)
    assert rd["count"] == int(g["count"]) and rd["count"] >= 1
    for f in range(rd["count"]):
        for k in ("rays_fg", "rays_bg", "depth_fg", "depth_bg", "T_wc"):
            assert np.array_equal(rd[k][f].numpy(), g[f"{k}_{f}"]), (k, f)
        assert np.array_equal(rd["pix_fg"][f], g[f"pix_fg_{f}"]) and np.array_equal(rd["pix_bg"][f], g[f"pix_bg_{f}"])


def test_metric_classes_follow_reference_definitions():
    from hortimapping_amd.metrics import ChamferDistance, PrecisionRecall
    A = np.array([[0.0, 0, 0], [0.004, 0, 0], [0.02, 0, 0]])
    Bp = np.array([[0.0, 0, 0.001], [0.02, 0.003, 0]])
    cd = ChamferDistance()
    cd.update(A, Bp)
    d_ab = np.array([0.001, np.sqrt(0.004 ** 2 + 0.001 ** 2), 0.003])
    d_ba = np.array([0.001, 0.003])
    assert abs(cd.compute() - 0.5 * (d_ab.mean() + d_ba.mean())) < 1e-12     # chamfer_distance.py:23-25 (gt=A, pt=B)
    pr = PrecisionRecall(0.001, 0.01, 100)
    pr.update(A, Bp)
    p, r, f, t = pr.compute_at_threshold(0.005)
    assert abs(t - 0.005) < 1e-4
    assert p == 100.0 and abs(r - 100.0) < 1e-9 and abs(f - 100.0) < 1e-9
    p, r, f, t = pr.compute_at_threshold(0.002)
    assert abs(p - 50.0) < 1e-9 and abs(r - 100 / 3) < 1e-9                   # precision: pred -> gt, recall: gt -> pred
    cd.update(A, np.zeros((0, 3)))                                             # empty prediction counts as 0 (:17-19)
    assert cd.cd_array[-1] == 0


def test_ply_roundtrip_and_mesh_helpers(tmp_path):
    from hortimapping_amd.mesher import TriangleMesh, read_ply, weld, write_ply
    soup = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1, 0, 0], [1, 1, 0], [0, 1, 0]]], dtype=np.float32)
    v, f = weld(soup)
    assert v.shape == (4, 3) and f.shape == (2, 3)
    m = TriangleMesh(v, f)
    assert abs(m.area() - 1.0) < 1e-6
    p = str(tmp_path / "m.ply")
    write_ply(m, p)
    m2 = read_ply(p)
    assert np.array_equal(m2.vertices, v) and np.array_equal(m2.faces, f)
    T = np.eye(4); T[:3, :3] *= 2; T[:3, 3] = [1, 2, 3]
    assert abs(m.transform(T).area() - 4.0) < 1e-5
    pts = m.sample_points_uniformly(1000)
    assert pts.shape == (1000, 3) and pts.min() >= 0 and pts.max() <= 1 and np.allclose(pts[:, 2], 0)


def test_pose_init_and_cleaning():
    from hortimapping_amd.data_prep import clean_pcd, get_pose_init, init_T_wo
    rs = np.random.RandomState(0)
    d = rs.randn(12000, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d[d[:, 2] < -0.3][:2000]                                              # camera-facing cap, like a real submap
    fruit = np.array([0.1, 0.2, 0.5]) + 0.04 * d
    noise = np.array([0.3, 0.3, 0.8]) + 0.002 * rs.randn(30, 3)               # an isolated small cluster
    clean = clean_pcd(np.concatenate([fruit, noise]), 0.01, 0.02)
    assert clean.shape[0] == 2000
    bg = np.array([0.1, 0.2 + 0.05, 0.5 + 0.06]) + 0.005 * rs.randn(200, 3)    # peduncle support behind/above
    c, rot, size, ok = get_pose_init(clean, bg)
    assert ok and 0.07 < size < 0.095 and abs(c[0] - 0.1) < 2e-3 and -0.8 < rot < 0.8
    opt = {"pose_init": {"rot_on": True, "scale_on": True}}
    T = init_T_wo(c, rot, size, opt, 0.08)
    assert abs(np.cbrt(np.linalg.det(T[:3, :3])) - max(size / (2 * 0.064), 0.5)) < 1e-9
    _, _, _, ok2 = get_pose_init(fruit * 10, bg)                                # 0.8 m object: rejected (utils.py:431-433)
    assert not ok2


def test_depth_filters_of_the_challenge_loader():
    """The two OpenCV calls of the challenge data loader (dataloader.py:50-53,66-71) as restated in datasets.py, against
    their definitions written out pixel by pixel: cv2.erode with an 11 x 11 rectangle = minimum over the in-image part of
    the window; cv2.bilateralFilter(d=3): radius-1 circular neighbourhood = the 5-point cross, Gaussian weights in range
    (sigma 15) and space (sigma 15), reflect-101 border.  (OpenCV itself is not in this image: cv2's float path
    evaluates the range weight through an interpolated table, so its output agrees with the formula to ~1e-4 relative.)"""
    from hortimapping_amd.datasets import _bilateral_3, _erode_11
    rs = np.random.RandomState(0)
    d = (0.3 + 0.7 * rs.rand(23, 31)).astype(np.float32)
    d[rs.rand(23, 31) < 0.1] = 0.0
    e = _erode_11(d)
    for (y, x) in [(0, 0), (5, 7), (22, 30), (11, 0), (0, 15), (12, 16)]:
        assert e[y, x] == d[max(0, y - 5):y + 6, max(0, x - 5):x + 6].min()
    assert e.dtype == d.dtype and e.shape == d.shape
    f = _bilateral_3(d)
    idx = lambda i, n: -i if i < 0 else (2 * n - 2 - i if i >= n else i)              # reflect-101
    for (y, x) in [(0, 0), (5, 7), (22, 30), (11, 0), (0, 15)]:
        num = den = 0.0
        for dy, dx in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)):
            v = float(d[idx(y + dy, 23), idx(x + dx, 31)])
            w = np.exp(-(v - float(d[y, x])) ** 2 / (2 * 15.0 ** 2)) * np.exp(-(dy * dy + dx * dx) / (2 * 15.0 ** 2))
            num += w * v
            den += w
        assert abs(f[y, x] - num / den) < 1e-6
    c = np.full((9, 9), 0.42, np.float32)
    assert np.allclose(_bilateral_3(c), c, atol=1e-7) and np.array_equal(_erode_11(c), c)


def test_voxel_down_sample_equals_rowwise_formulation():
    """Open3D-style voxel down-sampling of the Background submap (test_wild_completion.py:149-150): one point per occupied
    voxel = mean of its points, voxels in (ix, iy, iz) order, grid anchored at min_bound - voxel / 2.  The fast
    formulation (linear keys + bincount) gives the same bits as the plain one (np.unique over index rows + np.add.at)."""
    from hortimapping_amd.data_prep import voxel_down_sample
    rs = np.random.RandomState(4)
    pts = np.concatenate([rs.uniform(-0.3, 0.4, (20000, 3)), rs.uniform(-0.3, -0.29, (500, 3))])
    vs = 0.02
    origin = pts.min(axis=0) - 0.5 * vs
    idx = np.floor((pts - origin) / vs).astype(np.int64)
    _, inv, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    ref = np.zeros((len(cnt), 3))
    np.add.at(ref, inv.reshape(-1), pts)
    ref /= cnt[:, None]
    out = voxel_down_sample(pts, vs)
    assert out.shape == ref.shape and np.array_equal(out, ref) and cnt.max() > 3
    assert voxel_down_sample(np.zeros((0, 3)), vs).shape == (0, 3)
