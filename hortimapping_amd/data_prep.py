"""Caller-side data preparation (SURVEY.md 8f "next" rows 2 and 4): host code, once per instance, numpy only.

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


def clean_pcd(points: np.ndarray, cluster_dist_thre=0.01, outlier_point_ratio=0.02) -> np.ndarray:
    """`utils.py:407-417`: keep the most populated DBSCAN cluster."""
    from sklearn.cluster import DBSCAN
    n = points.shape[0]
    min_pts = max(1, int(n * outlier_point_ratio))
    labels = DBSCAN(eps=cluster_dist_thre, min_samples=min_pts).fit(points).labels_
    mode_label = Counter(labels.tolist()).most_common(1)[0][0]
    return points[labels == mode_label]


def clean_mesh(mesh, sample_point_count=5000, cluster_dist_thre=0.01, outlier_point_ratio=0.02, seed=0) -> np.ndarray:
    """`utils.py:389-405`: uniform surface samples of the submap mesh, then `clean_pcd`."""
    pts = mesh.sample_points_uniformly(sample_point_count, seed=seed)
    return clean_pcd(pts, cluster_dist_thre, outlier_point_ratio)


def get_pose_init(cur_points: np.ndarray, bg_points: np.ndarray, bbx_pad=0.01, min_bbx_size=0.03, max_bbx_size=0.16,
                  min_nearby_bg_pts=10, max_init_rot_deg=45):
    """`utils.py:420-459`: bbox centre (shifted along y), y-rotation from the peduncle support, bbox size, validity."""
    lo, hi = cur_points.min(axis=0), cur_points.max(axis=0)
    center, extent = 0.5 * (lo + hi), hi - lo
    bbx_size = float(extent.max() + bbx_pad)
    valid = not (bbx_size > max_bbx_size or bbx_size < min_bbx_size)
    rot_y = 0.0
    max_rot = max_init_rot_deg / 180.0 * math.pi
    if valid:
        center = center.copy()
        center[1] += (bbx_size - extent[1]) * 0.5
        if extent[1] == extent.max():
            center[1] += 0.01
        bmin = np.array([center[0] - 0.6 * bbx_size, center[1] - 0.8 * bbx_size, center[2] + 0.2 * bbx_size])
        bmax = np.array([center[0] + 0.6 * bbx_size, center[1] + 1.0 * bbx_size, center[2] + 1.2 * bbx_size])
        if bg_points is not None and len(bg_points):
            crop = bg_points[np.all((bg_points >= bmin) & (bg_points <= bmax), axis=1)]
            if len(crop) > min_nearby_bg_pts:
                rot_vec = np.mean(crop - center, axis=0)
                rot_y = 0.5 * math.pi - np.arctan2(rot_vec[2], rot_vec[0])
                rot_y = max(min(rot_y, max_rot), -max_rot)
    return center, float(rot_y), bbx_size, valid


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
    _, inv_, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    out = np.zeros((len(cnt), 3))
    np.add.at(out, inv_.reshape(-1), points)
    return out / cnt[:, None]
