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
