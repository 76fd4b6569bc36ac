"""Caller-side data preparation (SURVEY.md 8f "next" rows 2 and 4).  `get_render_data` is the reference's function on
the host (numpy); `get_render_data_device` produces the same dicts for MANY instances at once with the image scans
on the GPU (`csrc/hm_prep.hip`) -- at BUP20 sizes the host version costs ~0.2 s per instance, the optimisation 15 ms.

Mirrors of `wild_completion/utils.py`: `get_render_data` (:39-109), `clean_pcd` (:407-417, DBSCAN main cluster; Open3D's
cluster_dbscan replaced by scikit-learn's DBSCAN with the same eps / min_points), `get_pose_init` (:420-459, on plain
arrays instead of Open3D point clouds).  `get_rays` lives in `utils.py`.  These reproduce the reference's RNG use
(`np.random.choice` under the global numpy seed, utils.py:79,90) so that the same pixels are sampled."""
from __future__ import annotations

import math
from collections import Counter

import numpy as np
import torch

from .utils import get_rays


def get_render_data(submap_id, id_imgs, depth_imgs, cam_poses, img_size, invK, cfg, min_pix_count_match=400,
                    max_bbx_size=300, down_rate=1):
    """`utils.py:39-109`.  Returns the dict of per-frame lists the optimiser consumes (+ frame ids / pixels / count)."""
    render_data = {"frame_id": [], "T_wc": [], "rays_fg": [], "rays_bg": [], "depth_fg": [], "depth_bg": [],
                   "pix_fg": [], "pix_bg": [], "count": 0}
    cr = cfg["opt"]["render"]
    fg_pix_count, bg_pix_count, bg_pad = cr["n_fg_pix"], cr["n_bg_pix"], cr["n_bg_pad"]
    f32 = torch.float32
    for img_id, submap_id_img in id_imgs.items():
        depth_img = depth_imgs[img_id]
        mask_bool = submap_id_img == submap_id
        valid_mask_bool = mask_bool & (depth_img > 0.)
        if int(valid_mask_bool.sum()) < min_pix_count_match:                       # :56-58
            continue
        mask_v, mask_u = np.where(valid_mask_bool)
        min_v = max(int(mask_v.min()) - bg_pad, 0)
        max_v = min(int(mask_v.max()) + bg_pad, img_size[0] - 1)
        min_u = max(int(mask_u.min()) - bg_pad, 0)
        max_u = min(int(mask_u.max()) + bg_pad, img_size[1] - 1)
        bbx_h, bbx_w = max_v - min_v + 1, max_u - min_u + 1
        if bbx_h > max_bbx_size or bbx_w > max_bbx_size:                           # :65-67
            continue
        hh = np.linspace(min_v, max_v, int(bbx_h / down_rate)).astype(np.int32)
        ww = np.linspace(min_u, max_u, int(bbx_w / down_rate)).astype(np.int32)
        vv = np.repeat(hh, ww.shape[0])
        uu = np.tile(ww, hh.shape[0])
        valid_bg = ~mask_bool[vv, uu]
        pix_bg = np.stack([uu[valid_bg], vv[valid_bg]], axis=-1)                   # (u, v)
        depth_bg = depth_img[vv[valid_bg], uu[valid_bg]]
        if pix_bg.shape[0] > bg_pix_count:                                         # :78-82
            ind = np.random.choice(pix_bg.shape[0], bg_pix_count, replace=False)
            pix_bg, depth_bg = pix_bg[ind, :], depth_bg[ind]
        rays_bg = get_rays(pix_bg, invK).astype(np.float32)
        valid_fg = valid_mask_bool[vv, uu]
        pix_fg = np.stack([uu[valid_fg], vv[valid_fg]], axis=-1)
        depth_fg = depth_img[vv[valid_fg], uu[valid_fg]]
        if pix_fg.shape[0] > fg_pix_count:                                         # :89-93
            ind = np.random.choice(pix_fg.shape[0], fg_pix_count, replace=False)
            pix_fg, depth_fg = pix_fg[ind, :], depth_fg[ind]
        rays_fg = get_rays(pix_fg, invK).astype(np.float32)
        render_data["frame_id"].append(img_id)
        render_data["rays_fg"].append(torch.tensor(rays_fg, dtype=f32))
        render_data["rays_bg"].append(torch.tensor(rays_bg, dtype=f32))
        render_data["depth_fg"].append(torch.tensor(depth_fg, dtype=f32))
        render_data["depth_bg"].append(torch.tensor(depth_bg, dtype=f32))
        render_data["T_wc"].append(torch.tensor(cam_poses[img_id], dtype=f32))
        render_data["pix_fg"].append(pix_fg)
        render_data["pix_bg"].append(pix_bg)
        render_data["count"] += 1
    return render_data


class DeviceFrames:
    """The id / depth images of one sequence, uploaded once: id_imgs [F][H][W] int32, depth [F][H][W] float32 (cuda)."""

    def __init__(self, id_imgs: dict, depth_imgs: dict, device="cuda"):
        self.keys = list(id_imgs.keys())
        ids = np.stack([np.asarray(id_imgs[k]) for k in self.keys]).astype(np.int32)
        dep = np.stack([np.asarray(depth_imgs[k]) for k in self.keys]).astype(np.float32)
        self.F, self.H, self.W = ids.shape
        self.ids = torch.from_numpy(ids).to(device)
        self.depth = torch.from_numpy(dep).to(device)
        self.max_id = int(ids.max()) if ids.size else 0


def _prep_lib():
    import ctypes
    from . import _lib
    lib = _lib.lib()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.hm_prep_stats.restype = ci
    lib.hm_prep_stats.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, vp, vp]
    lib.hm_prep_scan.restype = ci
    lib.hm_prep_scan.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp]
    return lib


def get_render_data_device(submap_ids, frames: DeviceFrames, cam_poses, img_size, invK, cfg, min_pix_count_match=400,
                           max_bbx_size=300, down_rate=1):
    """`get_render_data` (`utils.py:39-109`) for every id of `submap_ids`, in that order: returns one dict per id with
    the same keys, dtypes and VALUES as successive host calls -- including the np.random.choice draws, which are made
    here on the host in the reference's order (instance by instance, frame by frame, background before foreground)
    from the candidate counts the device reports, so the global numpy RNG ends in the same state.  The device does the
    image scans: mask statistics of all instances in one pass over the sequence (`hm_prep_stats`), ordered candidate
    counting and rank-based gathering inside the padded boxes (`hm_prep_scan`), ray directions in fp64 like get_rays.
    Only `down_rate == 1` (the value every caller in the reference uses) is implemented on the device."""
    import ctypes
    from . import _lib
    if down_rate != 1:
        raise NotImplementedError("get_render_data_device: down_rate must be 1 (use get_render_data on the host)")
    lib = _prep_lib()
    dev = frames.ids.device
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cr = cfg["opt"]["render"]
    n_fg_want, n_bg_want, bg_pad = int(cr["n_fg_pix"]), int(cr["n_bg_pix"]), int(cr["n_bg_pad"])
    cap = max(n_fg_want, n_bg_want, 1)
    F, H, W = frames.F, frames.H, frames.W
    assert (H, W) == (int(img_size[0]), int(img_size[1])), "img_size does not match the images"
    B = len(submap_ids)
    out = [{"frame_id": [], "T_wc": [], "rays_fg": [], "rays_bg": [], "depth_fg": [], "depth_bg": [], "pix_fg": [],
            "pix_bg": [], "count": 0} for _ in range(B)]
    if B == 0 or F == 0:
        return out
    # pass 1: mask statistics of every wanted instance in every frame
    lut = np.full(max(frames.max_id, max(int(s) for s in submap_ids)) + 1, -1, np.int32)
    for slot, sid in enumerate(submap_ids):
        if int(sid) < 0:
            raise ValueError(f"submap id {int(sid)} is negative (ids index the instance-id images)")
        assert lut[int(sid)] < 0, "duplicate instance id"
        lut[int(sid)] = slot
    d_lut = torch.from_numpy(lut).to(dev)
    init = torch.tensor([0, 2 ** 31 - 1, -1, 2 ** 31 - 1, -1], dtype=torch.int32)
    d_stats = init.repeat(B * F).to(dev)
    _lib.check(lib.hm_prep_stats(frames.ids.data_ptr(), frames.depth.data_ptr(), F, H, W, d_lut.data_ptr(), len(lut), B,
                                 d_stats.data_ptr(), st), "hm_prep_stats")
    stats = d_stats.cpu().numpy().reshape(B, F, 5)
    # host: the reference's frame rejection rules and padded boxes (:56-67), pairs in the reference's loop order
    pairs = []
    for b in range(B):
        for f in range(F):
            cnt, v0, v1, u0, u1 = (int(x) for x in stats[b, f])
            if cnt < min_pix_count_match:
                continue
            min_v, max_v = max(v0 - bg_pad, 0), min(v1 + bg_pad, H - 1)
            min_u, max_u = max(u0 - bg_pad, 0), min(u1 + bg_pad, W - 1)
            if max_v - min_v + 1 > max_bbx_size or max_u - min_u + 1 > max_bbx_size:
                continue
            pairs.append([int(submap_ids[b]), f, min_v, max_v, min_u, max_u, -1, -1, b])
    P = len(pairs)
    if P == 0:
        return out
    pa = np.asarray(pairs, dtype=np.int32)
    d_pairs = torch.from_numpy(np.ascontiguousarray(pa[:, :8])).to(dev)
    d_counts = torch.zeros(P, 2, dtype=torch.int32, device=dev)
    nul = ctypes.c_void_p(0)
    _lib.check(lib.hm_prep_scan(frames.ids.data_ptr(), frames.depth.data_ptr(), H, W, d_pairs.data_ptr(), P, 0,
                                d_counts.data_ptr(), nul, nul, cap, nul, nul, nul, nul, st), "hm_prep_scan(count)")
    counts = d_counts.cpu().numpy()
    # host: the random draws, in the reference's order (:78-82 then :89-93 per pair)
    sel = np.zeros((P, 2, cap), np.int32)
    perm = np.zeros((P, 2, cap), np.int32)
    n_out = np.zeros((P, 2), np.int64)
    for p in range(P):
        for c, want in ((0, n_bg_want), (1, n_fg_want)):
            n = int(counts[p, c])
            if n > want:
                ind = np.random.choice(n, want, replace=False)
                order = np.argsort(ind, kind="stable")
                sel[p, c, :want] = ind[order]
                perm[p, c, :want] = order
                pa[p, 6 + c] = want
                n_out[p, c] = want
            else:
                n_out[p, c] = n                       # keep all, raster order
    d_pairs = torch.from_numpy(np.ascontiguousarray(pa[:, :8])).to(dev)
    d_sel, d_perm = torch.from_numpy(sel).to(dev), torch.from_numpy(perm).to(dev)
    d_invK = torch.from_numpy(np.ascontiguousarray(np.asarray(invK, dtype=np.float64).reshape(9))).to(dev)
    d_pix = torch.zeros(P, 2, cap, 2, dtype=torch.int32, device=dev)
    d_dep = torch.zeros(P, 2, cap, dtype=torch.float32, device=dev)
    d_rays = torch.zeros(P, 2, cap, 3, dtype=torch.float32, device=dev)
    _lib.check(lib.hm_prep_scan(frames.ids.data_ptr(), frames.depth.data_ptr(), H, W, d_pairs.data_ptr(), P, 1,
                                d_counts.data_ptr(), d_sel.data_ptr(), d_perm.data_ptr(), cap, d_invK.data_ptr(),
                                d_pix.data_ptr(), d_dep.data_ptr(), d_rays.data_ptr(), st), "hm_prep_scan(gather)")
    pix, dep, rays = d_pix.cpu().numpy(), d_dep.cpu(), d_rays.cpu()
    f32 = torch.float32
    for p in range(P):
        rd, f = out[int(pa[p, 8])], int(pa[p, 1])
        nb, nf = int(n_out[p, 0]), int(n_out[p, 1])
        img_id = frames.keys[f]
        rd["frame_id"].append(img_id)
        rd["rays_fg"].append(rays[p, 1, :nf].clone())
        rd["rays_bg"].append(rays[p, 0, :nb].clone())
        rd["depth_fg"].append(dep[p, 1, :nf].clone())
        rd["depth_bg"].append(dep[p, 0, :nb].clone())
        rd["T_wc"].append(torch.tensor(cam_poses[img_id], dtype=f32))
        rd["pix_fg"].append(pix[p, 1, :nf].copy())
        rd["pix_bg"].append(pix[p, 0, :nb].copy())
        rd["count"] += 1
    return out


def _mode_cluster(points, labels):
    mode_label = Counter(labels.tolist()).most_common(1)[0][0]
    return points[labels == mode_label]


def clean_pcd(points: np.ndarray, cluster_dist_thre=0.01, outlier_point_ratio=0.02) -> np.ndarray:
    """`utils.py:407-417`: keep the most populated DBSCAN cluster."""
    from sklearn.cluster import DBSCAN
    n = points.shape[0]
    min_pts = max(1, int(n * outlier_point_ratio))
    labels = DBSCAN(eps=cluster_dist_thre, min_samples=min_pts).fit(points).labels_
    return _mode_cluster(points, labels)


DBSCAN_MAX_POINTS = 5120


def dbscan_labels_device(clouds, eps, min_pts):
    """DBSCAN labels of several point clouds at once on the GPU (`hm_prep_dbscan`, one workgroup per cloud, at most
    DBSCAN_MAX_POINTS points each): same partition and same label numbering as scikit-learn's DBSCAN (clusters numbered
    by their first core point, border points to the earliest-numbered adjacent cluster, noise -1)."""
    import ctypes
    from . import _lib
    lib = _lib.lib()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.hm_prep_dbscan.restype = ci
    lib.hm_prep_dbscan.argtypes = [vp, vp, ci, ci, ctypes.c_double, vp, vp, vp]
    B = len(clouds)
    if B == 0:
        return []
    ns = [int(c.shape[0]) for c in clouds]
    n_stride = max(max(ns), 1)
    if n_stride > DBSCAN_MAX_POINTS:
        raise ValueError(f"dbscan_labels_device: at most {DBSCAN_MAX_POINTS} points per cloud (got {n_stride})")
    buf = np.zeros((B, n_stride, 3), np.float64)
    for b, c in enumerate(clouds):
        buf[b, :ns[b]] = c
    dev = torch.device("cuda")
    d_pts = torch.from_numpy(buf).to(dev)
    d_n = torch.tensor(ns, dtype=torch.int32, device=dev)
    d_mp = torch.tensor([int(m) for m in min_pts], dtype=torch.int32, device=dev)
    d_comp = torch.full((B, n_stride), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.hm_prep_dbscan(d_pts.data_ptr(), d_n.data_ptr(), n_stride, B, float(eps), d_mp.data_ptr(), d_comp.data_ptr(),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "hm_prep_dbscan")
    comp = d_comp.cpu().numpy()
    out = []
    for b in range(B):
        c = comp[b, :ns[b]]
        roots = np.unique(c[c >= 0])                      # ascending first-core index = scikit-learn's cluster order
        lab = np.full(ns[b], -1, np.int64)
        if len(roots):
            lab[c >= 0] = np.searchsorted(roots, c[c >= 0])
        out.append(lab)
    return out


def clean_pcd_device(clouds, cluster_dist_thre=0.01, outlier_point_ratio=0.02):
    """`clean_pcd` (`utils.py:407-417`) for a list of clouds, DBSCAN on the GPU.  A cloud beyond the kernel's
    DBSCAN_MAX_POINTS (its points and labels must fit one workgroup's LDS) takes the host `clean_pcd` instead -- same
    labels, see `dbscan_labels_device` -- and an empty cloud stays empty, so a configuration with a large
    `opt.recon.n_pts` keeps working."""
    out = [None] * len(clouds)
    on_dev = [i for i, c in enumerate(clouds) if 0 < c.shape[0] <= DBSCAN_MAX_POINTS]
    for i, c in enumerate(clouds):
        if c.shape[0] == 0:
            out[i] = c
        elif c.shape[0] > DBSCAN_MAX_POINTS:
            out[i] = clean_pcd(c, cluster_dist_thre, outlier_point_ratio)
    if on_dev:
        sub = [clouds[i] for i in on_dev]
        min_pts = [max(1, int(c.shape[0] * outlier_point_ratio)) for c in sub]
        for i, c, l in zip(on_dev, sub, dbscan_labels_device(sub, cluster_dist_thre, min_pts)):
            out[i] = _mode_cluster(c, l)
    return out


def clean_mesh(mesh, sample_point_count=5000, cluster_dist_thre=0.01, outlier_point_ratio=0.02, seed=0) -> np.ndarray:
    """`utils.py:389-405`: uniform surface samples of the submap mesh, then `clean_pcd`."""
    pts = mesh.sample_points_uniformly(sample_point_count, seed=seed)
    return clean_pcd(pts, cluster_dist_thre, outlier_point_ratio)


def pose_init_box(cur_points: np.ndarray, bbx_pad=0.01, min_bbx_size=0.03, max_bbx_size=0.16):
    """First half of `get_pose_init` (`utils.py:420-441`): bbox centre (shifted along y), bbox size, validity and the
    box [bmin, bmax] in which background points vote for the initial rotation."""
    lo, hi = cur_points.min(axis=0), cur_points.max(axis=0)
    center, extent = 0.5 * (lo + hi), hi - lo
    bbx_size = float(extent.max() + bbx_pad)
    valid = not (bbx_size > max_bbx_size or bbx_size < min_bbx_size)
    bmin = bmax = None
    if valid:
        center = center.copy()
        center[1] += (bbx_size - extent[1]) * 0.5
        if extent[1] == extent.max():
            center[1] += 0.01
        bmin = np.array([center[0] - 0.6 * bbx_size, center[1] - 0.8 * bbx_size, center[2] + 0.2 * bbx_size])
        bmax = np.array([center[0] + 0.6 * bbx_size, center[1] + 1.0 * bbx_size, center[2] + 1.2 * bbx_size])
    return center, bbx_size, valid, bmin, bmax


def pose_init_rotation(center, crop, min_nearby_bg_pts=10, max_init_rot_deg=45):
    """Second half (`utils.py:442-457`): y-rotation from the mean offset of the cropped background points."""
    rot_y = 0.0
    max_rot = max_init_rot_deg / 180.0 * math.pi
    if crop is not None and len(crop) > min_nearby_bg_pts:
        rot_vec = np.mean(crop - center, axis=0)
        rot_y = 0.5 * math.pi - np.arctan2(rot_vec[2], rot_vec[0])
        rot_y = max(min(rot_y, max_rot), -max_rot)
    return float(rot_y)


def get_pose_init(cur_points: np.ndarray, bg_points: np.ndarray, bbx_pad=0.01, min_bbx_size=0.03, max_bbx_size=0.16,
                  min_nearby_bg_pts=10, max_init_rot_deg=45):
    """`utils.py:420-459`: bbox centre (shifted along y), y-rotation from the peduncle support, bbox size, validity."""
    center, bbx_size, valid, bmin, bmax = pose_init_box(cur_points, bbx_pad, min_bbx_size, max_bbx_size)
    rot_y = 0.0
    if valid and bg_points is not None and len(bg_points):
        crop = bg_points[np.all((bg_points >= bmin) & (bg_points <= bmax), axis=1)]
        rot_y = pose_init_rotation(center, crop, min_nearby_bg_pts, max_init_rot_deg)
    return center, float(rot_y), bbx_size, valid


class DeviceCloud:
    """A point cloud (fp64) resident on the GPU, for repeated box crops (`crop_boxes`)."""

    def __init__(self, points: np.ndarray, device="cuda"):
        self.host = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        self.dev = torch.from_numpy(self.host).to(device)

    def __len__(self):
        return self.host.shape[0]

    def crop_boxes(self, bmins, bmaxs):
        """For every closed box [bmin, bmax]: the points inside, in input order -- what
        `pts[np.all((pts >= bmin) & (pts <= bmax), axis=1)]` returns (`hm_prep_box_select`: count pass, gather pass)."""
        import ctypes
        from . import _lib
        lib = _lib.lib()
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.hm_prep_box_select.restype = ci
        lib.hm_prep_box_select.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, vp]
        B = len(bmins)
        if B == 0:
            return []
        if len(self) == 0:
            return [self.host[:0] for _ in range(B)]
        dev = self.dev.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        boxes = torch.from_numpy(np.concatenate([np.asarray(bmins, np.float64).reshape(B, 3),
                                                 np.asarray(bmaxs, np.float64).reshape(B, 3)], axis=1)).to(dev)
        d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
        nul = ctypes.c_void_p(0)
        _lib.check(lib.hm_prep_box_select(self.dev.data_ptr(), len(self), boxes.data_ptr(), B, 0, d_counts.data_ptr(), nul,
                                          nul, st), "hm_prep_box_select(count)")
        counts = d_counts.cpu().numpy().astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(counts)])
        d_offs = torch.from_numpy(offs[:-1].copy()).to(dev)
        d_idx = torch.zeros(max(int(offs[-1]), 1), dtype=torch.int32, device=dev)
        _lib.check(lib.hm_prep_box_select(self.dev.data_ptr(), len(self), boxes.data_ptr(), B, 1, d_counts.data_ptr(),
                                          d_offs.data_ptr(), d_idx.data_ptr(), st), "hm_prep_box_select(gather)")
        idx = d_idx.cpu().numpy()
        return [self.host[idx[offs[b]:offs[b + 1]]] for b in range(B)]


def init_T_wo(center, rot_y_rad, bbx_size, cfg_opt, object_radius_max_m):
    """`test_wild_completion.py:196-209`: T_wo = [s R_y(theta) | centre]; s = max(bbx / (2 * 0.8 r_max), 0.5) when
    pose_init.scale_on, theta = 0 when pose_init.rot_on is false.  (The reference builds R with
    axis_angle_to_rotation_matrix, utils.py:371-378, which divides by the angle: theta == 0 is guarded here.)"""
    rot = rot_y_rad if cfg_opt["pose_init"]["rot_on"] else 0.0
    object_radius_m = object_radius_max_m * 0.8
    scale = max(bbx_size / (2.0 * object_radius_m), 0.5) if cfg_opt["pose_init"]["scale_on"] else 1.0
    c, s = math.cos(rot), math.sin(rot)
    T = np.eye(4)
    T[:3, :3] = scale * np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    T[:3, 3] = center
    return T


def final_pose_check(T_ow: np.ndarray, outlier_cfg: dict):
    """`test_wild_completion.py:228-246`: T_wo = inv(T_ow); scale = det(R s)^(1/3); zyx Euler angles of the rotation;
    an instance is dropped when the scale leaves [scale_min, scale_max] or |pitch| / |roll| exceed rot_max_deg.
    Returns (T_wo, final_scale, (yaw, pitch, roll) [deg], keep)."""
    from numpy.linalg import det, inv
    from scipy.spatial.transform import Rotation
    T_wo = inv(np.asarray(T_ow))
    final_scale = det(T_wo[:3, :3]) ** (1 / 3)
    yaw, pitch, roll = Rotation.from_matrix(T_wo[:3, :3] / final_scale).as_euler("zyx", degrees=True)
    keep = not (final_scale < outlier_cfg["scale_min"] or final_scale > outlier_cfg["scale_max"]
                or abs(pitch) > outlier_cfg["rot_max_deg"] or abs(roll) > outlier_cfg["rot_max_deg"])
    return T_wo, float(final_scale), (float(yaw), float(pitch), float(roll)), bool(keep)


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """Open3D `PointCloud.voxel_down_sample` (used on the Background submap, test_wild_completion.py:149-150): one
    point per occupied voxel = the mean of the points inside; the grid is anchored at min_bound - voxel_size / 2."""
    points = np.asarray(points, dtype=np.float64)
    if len(points) == 0:
        return points
    origin = points.min(axis=0) - 0.5 * voxel_size
    idx = np.floor((points - origin) / voxel_size).astype(np.int64)
    # one int64 key per voxel, ordered like the rows (ix, iy, iz) lexicographically; per-voxel sums by bincount, which adds
    # in input order exactly like np.add.at (same bits as the row-wise np.unique / np.add.at formulation, ~8x faster)
    dims = idx.max(axis=0) + 1
    key = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    _, inv_, cnt = np.unique(key, return_inverse=True, return_counts=True)
    inv_ = inv_.reshape(-1)
    out = np.stack([np.bincount(inv_, weights=points[:, c], minlength=len(cnt)) for c in range(3)], axis=1)
    return out / cnt[:, None]
