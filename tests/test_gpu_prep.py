"""Device-side `get_render_data` (csrc/hm_prep.hip) against the reference's own output (fixture G10,
captured from `wild_completion/utils.py:39-109`) and against the host mirror, which G10 pins: same frames kept, same
pixels in the same order, same depths, bit-identical rays, same state of the global numpy RNG afterwards."""
import time

import numpy as np
import pytest
import torch

from golden_util import load

pytestmark = pytest.mark.gpu


def _cfg(n_fg, n_bg, pad):
    return {"opt": {"render": {"n_fg_pix": n_fg, "n_bg_pix": n_bg, "n_bg_pad": pad}}}


def _same(rd_a, rd_b):
    assert rd_a["count"] == rd_b["count"] and rd_a["frame_id"] == rd_b["frame_id"]
    for f in range(rd_a["count"]):
        for k in ("rays_fg", "rays_bg", "depth_fg", "depth_bg", "T_wc"):
            a, b = rd_a[k][f], rd_b[k][f]
            assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), (k, f)
        for k in ("pix_fg", "pix_bg"):
            assert rd_a[k][f].dtype == rd_b[k][f].dtype and np.array_equal(rd_a[k][f], rd_b[k][f]), (k, f)


def test_device_render_prep_matches_reference_fixture():
    from hortimapping_amd.data_prep import DeviceFrames, get_render_data_device
    g = load("g10_data_prep")
    cfg = _cfg(int(g["n_fg_pix"]), int(g["n_bg_pix"]), int(g["n_bg_pad"]))
    idimg, depth = g["id_img"], g["depth_img"]
    frames = DeviceFrames({0: idimg, 3: idimg.T.copy()}, {0: depth, 3: depth.T.copy()})
    np.random.seed(42)
    rd = get_render_data_device([5], frames, {0: np.eye(4), 3: np.eye(4)}, (64, 64), np.linalg.inv(g["K64"]), cfg,
                                min_pix_count_match=100, max_bbx_size=300)[0]
    assert rd["count"] == int(g["count"]) and rd["count"] >= 1
    for f in range(rd["count"]):
        for k in ("rays_fg", "rays_bg", "depth_fg", "depth_bg", "T_wc"):
            assert np.array_equal(rd[k][f].numpy(), g[f"{k}_{f}"]), (k, f)
        assert np.array_equal(rd["pix_fg"][f], g[f"pix_fg_{f}"]) and np.array_equal(rd["pix_bg"][f], g[f"pix_bg_{f}"])


def _scene(rs, F, H, W, n_inst, r_lo, r_hi):
    yy, xx = np.mgrid[0:H, 0:W]
    ids, dep, pose = {}, {}, {}
    for f in range(F):
        img = np.zeros((H, W), np.int32)
        for k in range(n_inst):
            if rs.rand() < 0.15:
                continue                                       # instance not visible in this frame
            cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(r_lo, r_hi)   # may be clipped by the border
            img[(yy - cy) ** 2 + ((xx - cx) * 0.8) ** 2 < r * r] = k + 2
        d = (0.4 + 0.3 * rs.rand(H, W)).astype(np.float32)
        d[rs.rand(H, W) < 0.1] = 0.0                           # depth holes
        key = 10 * f + 1                                       # frame keys need not be 0..F-1
        ids[key], dep[key], pose[key] = img, d, np.eye(4) + 0.01 * rs.randn(4, 4)
    return ids, dep, pose


@pytest.mark.parametrize("case", ["small", "bup20_size"])
def test_device_render_prep_matches_host_mirror(case):
    from hortimapping_amd.data_prep import DeviceFrames, get_render_data, get_render_data_device
    rs = np.random.RandomState(3)
    if case == "small":
        F, H, W, n_inst, cfg, kw = 5, 96, 128, 6, _cfg(40, 30, 7), dict(min_pix_count_match=60, max_bbx_size=70)
        ids, dep, pose = _scene(rs, F, H, W, n_inst, 4, 30)    # some below the count threshold, some boxes too large
    else:
        F, H, W, n_inst, cfg, kw = 12, 720, 1280, 24, _cfg(200, 200, 20), dict(min_pix_count_match=400, max_bbx_size=300)
        ids, dep, pose = _scene(rs, F, H, W, n_inst, 10, 140)
    K = np.array([[612.3, 0.0, W / 2 + 0.37], [0.0, 609.1, H / 2 - 1.21], [0.0, 0.0, 1.0]])
    invK = np.linalg.inv(K)
    sids = [k + 2 for k in range(n_inst)][::-1] + [n_inst + 7]          # any order; one id that never appears
    np.random.seed(42)
    t0 = time.perf_counter()
    host = [get_render_data(s, ids, dep, pose, (H, W), invK, cfg, **kw) for s in sids]
    t_host = time.perf_counter() - t0
    state_host = np.random.get_state()
    frames = DeviceFrames(ids, dep)
    np.random.seed(42)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    devr = get_render_data_device(sids, frames, pose, (H, W), invK, cfg, **kw)
    t_dev = time.perf_counter() - t0
    state_dev = np.random.get_state()
    assert sum(r["count"] for r in host) > n_inst and any(r["count"] < F for r in host)
    for a, b in zip(devr, host):
        _same(a, b)
    assert state_host[0] == state_dev[0] and np.array_equal(state_host[1], state_dev[1]) and state_host[2:] == state_dev[2:]
    print(f"\n{case}: {len(sids)} instances x {F} frames of {H}x{W}: host {t_host * 1e3:.1f} ms, device {t_dev * 1e3:.1f} ms")


def test_device_render_prep_empty_inputs():
    from hortimapping_amd.data_prep import DeviceFrames, get_render_data_device
    ids = {0: np.zeros((32, 32), np.int32)}
    dep = {0: np.ones((32, 32), np.float32)}
    frames = DeviceFrames(ids, dep)
    out = get_render_data_device([3, 4], frames, {0: np.eye(4)}, (32, 32), np.eye(3), _cfg(10, 10, 2))
    assert [r["count"] for r in out] == [0, 0]
    assert get_render_data_device([], frames, {0: np.eye(4)}, (32, 32), np.eye(3), _cfg(10, 10, 2)) == []


def _fruit_cloud(rs, n, n_out_cluster, n_noise):
    d = rs.randn(6 * n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d[d[:, 2] < -0.2][:n]
    return np.concatenate([0.04 * d + [0.1, 0.2, 0.5], 0.004 * rs.randn(n_out_cluster, 3) + [0.17, 0.2, 0.5],
                           rs.uniform(-0.1, 0.1, (n_noise, 3)) + [0.1, 0.2, 0.5]])


def test_device_dbscan_equals_scikit_learn():
    """`hm_prep_dbscan` against scikit-learn's DBSCAN (the stand-in for Open3D's cluster_dbscan in clean_pcd,
    utils.py:407-417): identical label arrays -- same core points, same clusters in the same numbering, border points
    assigned to the same cluster, same noise -- on fruit-like caps with satellite clusters and noise, on touching
    blobs (border points adjacent to two clusters), on a cloud that is all noise and on one that is one cluster."""
    from sklearn.cluster import DBSCAN
    from hortimapping_amd.data_prep import clean_pcd, clean_pcd_device, dbscan_labels_device
    rs = np.random.RandomState(5)
    clouds = [_fruit_cloud(rs, 1800, 150, 50), _fruit_cloud(rs, 4600, 300, 100), _fruit_cloud(rs, 400, 0, 10)]
    a = rs.randn(600, 3) * 0.006
    clouds.append(np.concatenate([a, a + [0.024, 0, 0], a + [0.012, 0.021, 0]]))        # three touching blobs
    clouds.append(rs.uniform(-1, 1, (700, 3)))                                           # all noise at eps = 1 cm
    clouds.append(0.003 * rs.randn(900, 3))                                              # one cluster
    clouds.append(rs.uniform(0, 0.05, (5120, 3)))                                        # the size limit
    for eps, ratio in ((0.01, 0.02), (0.006, 0.01)):
        mp = [max(1, int(len(c) * ratio)) for c in clouds]
        got = dbscan_labels_device(clouds, eps, mp)
        for c, m, g in zip(clouds, mp, got):
            ref = DBSCAN(eps=eps, min_samples=m).fit(c).labels_
            assert np.array_equal(g, ref), (len(c), eps, m, int((g != ref).sum()))
    kept_dev = clean_pcd_device(clouds[:4], 0.01, 0.02)
    for c, k in zip(clouds[:4], kept_dev):
        assert np.array_equal(k, clean_pcd(c, 0.01, 0.02))
    with pytest.raises(ValueError):
        dbscan_labels_device([np.zeros((5121, 3))], 0.01, [1])


def test_device_box_crop_equals_numpy():
    """`hm_prep_box_select` (the background crop of get_pose_init, utils.py:442-447): same points in the same order as
    the numpy boolean mask, closed bounds included, empty and all-inclusive boxes; get_pose_init assembled from the
    two halves + the device crop equals the host function bit for bit."""
    from hortimapping_amd.data_prep import DeviceCloud, get_pose_init, pose_init_box, pose_init_rotation
    rs = np.random.RandomState(9)
    bg = rs.uniform(-0.5, 0.5, (300001, 3))
    bg[1000] = [0.1, 0.2, 0.3]
    cloud = DeviceCloud(bg)
    bmins = [np.array([0.1, 0.2, 0.3]), np.array([-0.2, -0.1, 0.0]), np.array([5.0, 5.0, 5.0]), np.array([-1.0, -1.0, -1.0])]
    bmaxs = [np.array([0.2, 0.3, 0.4]), np.array([0.1, 0.25, 0.3]), np.array([6.0, 6.0, 6.0]), np.array([1.0, 1.0, 1.0])]
    crops = cloud.crop_boxes(bmins, bmaxs)
    for lo, hi, c in zip(bmins, bmaxs, crops):
        ref = bg[np.all((bg >= lo) & (bg <= hi), axis=1)]
        assert c.shape == ref.shape and np.array_equal(c, ref)
    assert len(crops[2]) == 0 and len(crops[3]) == len(bg) and np.array_equal(crops[0][0], bg[1000])   # closed lower bound
    fruits = [_fruit_cloud(rs, 1500, 0, 0) + rs.uniform(-0.3, 0.3, 3) for _ in range(6)]
    boxes = [pose_init_box(f) for f in fruits]
    assert all(b[2] for b in boxes)
    crops = cloud.crop_boxes([b[3] for b in boxes], [b[4] for b in boxes])
    for f, (center, size, valid, _, _), crop in zip(fruits, boxes, crops):
        c_ref, rot_ref, size_ref, valid_ref = get_pose_init(f, bg)
        assert np.array_equal(center, c_ref) and size == size_ref and valid == valid_ref
        assert pose_init_rotation(center, crop) == rot_ref
    assert DeviceCloud(np.zeros((0, 3))).crop_boxes(bmins[:1], bmaxs[:1])[0].shape == (0, 3)


def test_randomised_sweep_of_the_device_kernels():
    """Random sizes and parameters for the four kernels of this row: render prep vs the host mirror (image sizes that are
    not multiples of anything, clipped discs, depth holes, thresholds that reject frames, RNG state), DBSCAN vs
    scikit-learn, marching cubes on random grids of random size (closed + consistently oriented), box crops vs numpy."""
    from hortimapping_amd import data_prep as DP
    from hortimapping_amd.mesher import extract_surface, weld
    from sklearn.cluster import DBSCAN
    rs = np.random.RandomState(123)
    bad = 0
    # 1. render prep on random sizes
    for trial in range(40):
        H, W = rs.randint(20, 200), rs.randint(20, 260)
        F, NI = rs.randint(1, 5), rs.randint(1, 7)
        yy, xx = np.mgrid[0:H, 0:W]
        ids, dep, pose = {}, {}, {}
        for f in range(F):
            img = np.zeros((H, W), np.int32)
            for k in range(NI):
                cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(2, max(3, min(H, W) // 2))
                img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = k + 1
            d = rs.rand(H, W).astype(np.float32); d[rs.rand(H, W) < 0.2] = 0
            ids[f * 3], dep[f * 3], pose[f * 3] = img, d, np.eye(4)
        cfg = {"opt": {"render": {"n_fg_pix": int(rs.randint(1, 60)), "n_bg_pix": int(rs.randint(1, 60)), "n_bg_pad": int(rs.randint(0, 12))}}}
        kw = dict(min_pix_count_match=int(rs.randint(1, 80)), max_bbx_size=int(rs.randint(10, 300)))
        invK = np.linalg.inv(np.array([[500 + rs.rand(), 0, W / 2], [0, 480 + rs.rand(), H / 2], [0, 0, 1]]))
        sids = list(rs.permutation(NI) + 1) + [NI + 5]
        np.random.seed(trial); host = [DP.get_render_data(s, ids, dep, pose, (H, W), invK, cfg, **kw) for s in sids]; st_h = np.random.get_state()[1].copy()
        np.random.seed(trial); devr = DP.get_render_data_device(sids, DP.DeviceFrames(ids, dep), pose, (H, W), invK, cfg, **kw); st_d = np.random.get_state()[1]
        ok = np.array_equal(st_h, st_d)
        for a, b in zip(devr, host):
            ok &= a["count"] == b["count"] and a["frame_id"] == b["frame_id"]
            for f in range(min(a["count"], b["count"])):
                for k in ("rays_fg", "rays_bg", "depth_fg", "depth_bg"):
                    ok &= a[k][f].shape == b[k][f].shape and torch.equal(a[k][f], b[k][f])
                ok &= np.array_equal(a["pix_fg"][f], b["pix_fg"][f]) and np.array_equal(a["pix_bg"][f], b["pix_bg"][f])
        if not ok: bad += 1; print("render prep mismatch in trial", trial, H, W, F, NI, cfg, kw)
    assert bad == 0, "render prep"
    # 2. DBSCAN random
    bad = 0
    for trial in range(40):
        n = rs.randint(1, 3000)
        k = rs.randint(1, 6)
        cen = rs.uniform(-0.1, 0.1, (k, 3))
        pts = cen[rs.randint(0, k, n)] + rs.uniform(0.001, 0.02) * rs.randn(n, 3)
        eps = float(rs.uniform(0.002, 0.03)); mp = int(rs.randint(1, 40))
        got = DP.dbscan_labels_device([pts], eps, [mp])[0]
        ref = DBSCAN(eps=eps, min_samples=mp).fit(pts).labels_
        if not np.array_equal(got, ref): bad += 1; print("dbscan mismatch", trial, n, eps, mp, int((got != ref).sum()))
    assert bad == 0, "dbscan"
    # 3. marching cubes random grids of random sizes: closed + oriented
    bad = 0
    for trial in range(20):
        n = rs.randint(3, 24)
        g = torch.from_numpy(rs.rand(2, n, n, n).astype(np.float32) - 0.5)
        g[:, 0] = g[:, -1] = 1; g[:, :, 0] = g[:, :, -1] = 1; g[:, :, :, 0] = g[:, :, :, -1] = 1
        for soup in extract_surface(g.cuda(), 1.0, method="mc"):
            v, f = weld(soup)
            if f.shape[0] == 0: continue
            de = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
            key = de[:, 0].astype(np.int64) * (v.shape[0] + 1) + de[:, 1]; rev = de[:, 1].astype(np.int64) * (v.shape[0] + 1) + de[:, 0]
            if not (np.unique(key).shape[0] == key.shape[0] and np.array_equal(np.sort(key), np.sort(rev))): bad += 1; print("mc non-manifold", trial, n)
    assert bad == 0, "marching cubes"
    # 4. box crop random
    bad = 0
    for trial in range(20):
        n = rs.randint(0, 50000); pts = rs.uniform(-1, 1, (n, 3)); B = rs.randint(1, 9)
        lo = rs.uniform(-1, 0.5, (B, 3)); hi = lo + rs.uniform(0, 1.5, (B, 3))
        crops = DP.DeviceCloud(pts).crop_boxes(list(lo), list(hi))
        for b in range(B):
            ref = pts[np.all((pts >= lo[b]) & (pts <= hi[b]), axis=1)]
            if not (crops[b].shape == ref.shape and np.array_equal(crops[b], ref)): bad += 1; print("crop mismatch", trial, b)
    assert bad == 0, "box crop"
