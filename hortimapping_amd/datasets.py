"""On-disk data formats of the two reference entry points (SURVEY.md 8f "next" row 4), without Open3D / OpenCV /
scikit-image (absent from the image): readers for the BUP20 "wild" layout (`test_wild_completion.py:72-122,133-151`)
and for the shape-completion-challenge layout (`dataloader.py:9-153`), plus writers that lay SYNTHETIC fruit scenes
out in exactly those layouts (the real datasets are downloads the reference fetches with wget; no network here).

    BUP20:      <data_dir>/<frame>_submap_id.png | _depth.tiff | _color.png | _pose.txt (16 floats, row-major T_wc)
                <data_dir>/submaps/<id>_<Category>.ply        cam_info.yaml {intrinsics, extrinsics, img_size [H, W]}
    challenge:  <data_dir>/<split>/<fid>/input/{intrinsic.json (column-major 3x3), masks/*.png, poses/*.txt,
                color/*.png, depth/*.npy}, gt/pcd/fruit.ply
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List

import numpy as np
import yaml
from PIL import Image

from . import synthetic as S
from .ply import TriangleMesh, read_ply, write_ply


# ------------------------------------------------------------------------------------------------ point-cloud PLY
def write_points_ply(points: np.ndarray, path: str):
    v = np.ascontiguousarray(points, dtype="<f4")
    header = ("ply\nformat binary_little_endian 1.0\n" f"element vertex {v.shape[0]}\n"
              "property float x\nproperty float y\nproperty float z\nend_header\n")
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(v.tobytes())


def read_points_ply(path: str) -> np.ndarray:
    return read_ply(path).vertices


# ------------------------------------------------------------------------------------------------ BUP20 reader
def load_cam_info(path: str):
    cam = yaml.safe_load(open(path))                                     # test_wild_completion.py:72-77
    return np.array(cam["intrinsics"], dtype=np.float64), np.array(cam["extrinsics"], dtype=np.float64), cam["img_size"]


def load_bup20_frames(cfg) -> Dict[str, dict]:
    """test_wild_completion.py:86-122: every `*id*` file with its depth / colour / pose siblings, subject to
    begin_frame / end_frame / every_frame.  Returns the four dictionaries keyed by frame id."""
    base = cfg["data_dir"]
    id_imgs, depth_imgs, rgb_imgs, cam_poses = {}, {}, {}, {}
    frame_count = 0
    for fname in sorted(os.listdir(base)):
        if "id" not in fname:
            continue
        if frame_count < cfg["begin_frame"] or frame_count > cfg["end_frame"] or frame_count % cfg["every_frame"] != 0:
            frame_count += 1
            continue
        p = os.path.join(base, fname)
        frame_id = fname.split("_")[0]
        id_imgs[frame_id] = np.array(Image.open(p))
        depth_imgs[frame_id] = np.array(Image.open(p.replace("submap_id.png", "depth.tiff"))).astype(float)
        rgb_imgs[frame_id] = np.array(Image.open(p.replace("submap_id.png", "color.png"))).astype(float)
        pose_path = p.replace("submap_id.png", "pose.txt")
        T = np.eye(4)
        if os.path.isfile(pose_path):
            T = np.array([float(x) for x in open(pose_path).read().split()]).reshape(4, 4)
        cam_poses[frame_id] = T
        frame_count += 1
    return {"id": id_imgs, "depth": depth_imgs, "rgb": rgb_imgs, "pose": cam_poses}


# ------------------------------------------------------------------------------------------------ challenge reader
def _bilateral_3(depth: np.ndarray, sigma_c=15.0, sigma_s=15.0) -> np.ndarray:
    """cv2.bilateralFilter(depth, 3, 15, 15) (dataloader.py:66-68): radius-1 (cross-shaped) neighbourhood,
    reflect-101 border.  All in fp32 with in-place temporaries (round 6: the fp64 temporaries of the first version -- a
    np.float64 spatial weight promoted every product -- made this filter 0.8 of the 1.07 s the challenge script spent
    reading its 64 x 5 frames); results differ from that version at the 1e-7 level, inside the caveat this restatement
    carries anyway (OpenCV is not in the image: DESIGN.md section 7)."""
    d = depth.astype(np.float32)
    p = np.pad(d, 1, mode="reflect")
    k_c = np.float32(-1.0 / (2.0 * sigma_c ** 2))
    w_s = np.float32(np.exp(-1.0 / (2.0 * sigma_s ** 2)))        # the four neighbours are at distance 1
    acc = d.copy()                                                # centre pixel: weight exactly 1
    wsum = np.ones_like(d)
    t = np.empty_like(d)
    for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        nb = p[1 + dy:1 + dy + d.shape[0], 1 + dx:1 + dx + d.shape[1]]
        np.subtract(nb, d, out=t)
        np.multiply(t, t, out=t)
        t *= k_c
        np.exp(t, out=t)
        t *= w_s
        wsum += t
        t *= nb
        acc += t
    acc /= wsum
    return acc


def _erode_11(depth: np.ndarray) -> np.ndarray:
    """cv2.erode with the 11x11 rectangular element (dataloader.py:50-53,71): minimum filter."""
    from scipy.ndimage import minimum_filter
    return minimum_filter(depth, size=11, mode="constant", cval=np.inf).astype(depth.dtype)


class ShapeCompletionDataset:
    """Mirror of `dataloader.ShapeCompletionDataset` (dataloader.py:9-153); point clouds are (N,3) arrays."""

    def __init__(self, data_source=None, split="train", return_pcd=True, return_rgbd=True):
        assert return_pcd or return_rgbd
        self.data_source, self.split = data_source, split
        self.return_pcd, self.return_rgbd = return_pcd, return_rgbd
        root = os.path.join(data_source, split)
        self.fruit_list = {fid: {"path": os.path.join(root, fid)} for fid in os.listdir(root)}   # unsorted, :27-33

    @staticmethod
    def load_K(path):
        data = json.load(open(path))["intrinsic_matrix"]
        return np.reshape(data, (3, 3), order="F")                       # :100-104 column-major

    @staticmethod
    def rgbd_to_pcd(depth, mask, pose, K):
        """:107-128: back-project depth*mask (truncated at 1 m) and move to the world with `pose` (T_wc)."""
        d = depth * (mask > 0)
        v, u = np.nonzero((d > 0) & (d < 1.0))
        z = d[v, u].astype(np.float64)
        pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
        return pc @ pose[:3, :3].T + pose[:3, 3]

    def get_rgbd(self, fid):
        root = self.fruit_list[fid]["path"]
        K = self.load_K(os.path.join(root, "input/intrinsic.json"))
        out = {"intrinsic": K, "pcd": np.zeros((0, 3)), "frames": {}}
        for frameid in os.listdir(os.path.join(root, "input/masks/")):
            pose = np.loadtxt(os.path.join(root, "input/poses/", frameid.replace("png", "txt")))
            rgb = np.array(Image.open(os.path.join(root, "input/color/", frameid)).convert("RGB"))
            depth = np.load(os.path.join(root, "input/depth/", frameid.replace("png", "npy")))
            depth = _erode_11(_bilateral_3(depth))                       # :66-71
            mask = np.array(Image.open(os.path.join(root, "input/masks/", frameid)).convert("L"))
            key = frameid.replace(".png", "")
            out["frames"][key] = {"rgb": rgb, "depth": depth, "mask": mask, "pose": pose, "fname": key}
            if self.return_pcd:
                out["pcd"] = np.concatenate([out["pcd"], self.rgbd_to_pcd(depth, mask, pose, K)], axis=0)
        return out

    def __len__(self):
        return len(self.fruit_list)

    def __getitem__(self, idx):
        fid = list(self.fruit_list.keys())[idx]
        item = {}
        if self.split != "test":
            item["groundtruth_pcd"] = read_points_ply(os.path.join(self.fruit_list[fid]["path"], "gt/pcd/fruit.ply"))
        data = self.get_rgbd(fid)
        if self.return_pcd:
            item["rgbd_pcd"] = data["pcd"]
        if self.return_rgbd:
            item["rgbd_intrinsic"] = data["intrinsic"]
            item["rgbd_frames"] = data["frames"]
        item["fid"] = fid
        return item

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


# ------------------------------------------------------------------------------------------------ synthetic scenes
def _render_fruit(sdf_world: Callable, cam_T_wc: np.ndarray, K: np.ndarray, img_size, centre, r_max):
    """Depth (z along the optical axis) of one fruit in one camera; 0 where the pixel misses it."""
    H, W = img_size
    T_cw = np.linalg.inv(cam_T_wc)
    pc = T_cw[:3, :3] @ centre + T_cw[:3, 3]
    uc, vc = K[0, 0] * pc[0] / pc[2] + K[0, 2], K[1, 1] * pc[1] / pc[2] + K[1, 2]
    rad = K[0, 0] * r_max / pc[2] * 1.2
    u0, u1 = max(0, int(uc - rad)), min(W - 1, int(uc + rad))
    v0, v1 = max(0, int(vc - rad)), min(H - 1, int(vc + rad))
    depth = np.zeros((H, W), dtype=np.float32)
    if u1 <= u0 or v1 <= v0:
        return depth
    uu, vv = np.meshgrid(np.arange(u0, u1 + 1), np.arange(v0, v1 + 1))
    uv1 = np.stack([uu.ravel(), vv.ravel(), np.ones(uu.size)], axis=1).astype(np.float64)
    dirs_c = uv1 @ np.linalg.inv(K).T                                # z = 1
    dirs_w = dirs_c @ cam_T_wc[:3, :3].T
    org = np.broadcast_to(cam_T_wc[:3, 3], dirs_w.shape).copy()
    hit, t = S._first_hit(sdf_world, org, dirs_w, pc[2] - r_max, pc[2] + r_max, n_march=40, n_bisect=14)
    depth[vv.ravel()[hit], uu.ravel()[hit]] = t[hit]
    return depth


def _depth_mesh(depth: np.ndarray, mask: np.ndarray, K: np.ndarray, T_wc: np.ndarray) -> TriangleMesh:
    """Triangle mesh of the back-projected masked depth pixels (two triangles per 2x2 pixel block inside the mask)."""
    H, W = depth.shape
    idx = -np.ones((H, W), dtype=np.int64)
    v, u = np.nonzero(mask)
    idx[v, u] = np.arange(v.size)
    z = depth[v, u].astype(np.float64)
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    pw = pc @ T_wc[:3, :3].T + T_wc[:3, 3]
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    ok = (a >= 0) & (b >= 0) & (c >= 0) & (d >= 0)
    f = np.concatenate([np.stack([a[ok], c[ok], b[ok]], 1), np.stack([b[ok], c[ok], d[ok]], 1)], 0)
    return TriangleMesh(pw.astype(np.float32), f.astype(np.int32))


def synthetic_scene(params, sdf_fn_factory, n_fruits=3, seed=7, r_max=0.08):
    """Fruit centres / true latents / world-frame sdf callables of a small synthetic plant."""
    rs = np.random.RandomState(seed)
    L = int(params["latent_dim"])
    Ws, bs = S.fold_weight_norm(params)
    fruits = []
    cols = n_fruits if n_fruits <= 6 else int(np.ceil(np.sqrt(n_fruits * 1.5)))     # a row, or a cols x rows wall of fruits
    rows = (n_fruits + cols - 1) // cols
    for i in range(n_fruits):
        z_true = (0.07 * rs.randn(L)).astype(np.float32)
        if rows == 1:
            centre = np.array([0.16 * (i - (n_fruits - 1) / 2), 0.02 * rs.randn(), 0.5 + 0.03 * rs.randn()])
        else:                  # 12 cm pitch at 0.6 m: every fruit stays inside a 720 x 1280 image of the f = 300 px camera
            cx, cy = i % cols, i // cols
            centre = np.array([0.12 * (cx - (cols - 1) / 2) + 0.005 * rs.randn(), 0.12 * (cy - (rows - 1) / 2) + 0.005 * rs.randn(),
                               0.6 + 0.02 * rs.randn()])
        f_obj = sdf_fn_factory(z_true) if sdf_fn_factory else (lambda p, z=z_true: S.np_decoder_forward(Ws, bs, z, p))
        fruits.append({"z_true": z_true, "centre": centre,
                       "sdf_world": (lambda p, c=centre, f=f_obj: f(p - c))})
    return fruits


def write_synthetic_bup20(root: str, params, sdf_fn_factory=None, n_fruits=3, n_frames=4, img_size=(240, 320),
                          seed=7) -> dict:
    """Lay a synthetic scene out in the BUP20 folder format consumed by test_wild_completion.py (C1 plumbing)."""
    os.makedirs(os.path.join(root, "submaps"), exist_ok=True)
    H, W = img_size
    K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1.0]])
    yaml.safe_dump({"intrinsics": K.tolist(), "extrinsics": np.eye(4).tolist(), "img_size": [H, W]},
                   open(os.path.join(root, "cam_info.yaml"), "w"))
    fruits = synthetic_scene(params, sdf_fn_factory, n_fruits, seed)
    bg_depth = 0.9
    meshes: Dict[int, List[TriangleMesh]] = {i: [] for i in range(n_fruits)}
    for f in range(n_frames):
        T_wc = np.eye(4)
        T_wc[:3, 3] = [0.03 * (f - (n_frames - 1) / 2), 0.0, 0.0]
        idimg = np.zeros((H, W), dtype=np.uint8)
        depth = np.full((H, W), bg_depth, dtype=np.float32)
        for i, fr in enumerate(fruits):
            d = _render_fruit(fr["sdf_world"], T_wc, K, img_size, fr["centre"], 0.08)
            m = (d > 0) & (d < depth)
            depth[m] = d[m]
            idimg[m] = i + 2                                            # submap ids 2.. (1 = Background)
        depth[5:9, 5:9] = 0.0                                           # a few invalid-depth pixels
        name = f"{f:06d}"
        Image.fromarray(idimg).save(os.path.join(root, f"{name}_submap_id.png"))
        Image.fromarray(depth, mode="F").save(os.path.join(root, f"{name}_depth.tiff"))
        col = np.stack([60 + 40 * (idimg > 0), 120 + 10 * idimg, 60 + 0 * idimg], axis=-1).astype(np.uint8)
        Image.fromarray(col).save(os.path.join(root, f"{name}_color.png"))
        open(os.path.join(root, f"{name}_pose.txt"), "w").write(" ".join(f"{x:.9f}" for x in T_wc.reshape(-1)))
        if f in (0, n_frames - 1):
            for i in range(n_fruits):
                meshes[i].append(_depth_mesh(depth, idimg == i + 2, K, T_wc))
    # background submap: the back wall plus a small blob above every fruit (the peduncle support get_pose_init uses)
    gx, gy = np.meshgrid(np.linspace(-0.5, 0.5, 60), np.linspace(-0.4, 0.4, 50))
    wall = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, bg_depth)], 1)
    idx = np.arange(gx.size).reshape(gx.shape)
    wf = np.concatenate([np.stack([idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel()], 1),
                         np.stack([idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()], 1)], 0)
    write_ply(TriangleMesh(wall.astype(np.float32), wf.astype(np.int32)), os.path.join(root, "submaps", "1_Background.ply"))
    for i in range(n_fruits):
        v = np.concatenate([m.vertices for m in meshes[i]], 0)
        off, fs = 0, []
        for m in meshes[i]:
            fs.append(m.faces + off)
            off += m.vertices.shape[0]
        write_ply(TriangleMesh(v, np.concatenate(fs, 0)), os.path.join(root, "submaps", f"{i + 2}_SweetPepper.ply"))
    return {"K": K, "fruits": fruits}


def write_synthetic_challenge(root: str, split: str, params, sdf_fn_factory=None, n_fruits=3, n_frames=5,
                              img_size=(240, 320), seed=11) -> dict:
    """Lay synthetic fruits out in the shape-completion-challenge folder format (C3 plumbing).  The fruit frame is the
    world frame (the challenge gives poses relative to the fruit, run_shape_completion_challenge.py:207-209)."""
    H, W = img_size
    K = np.array([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1.0]])
    fruits = synthetic_scene(params, sdf_fn_factory, n_fruits, seed)
    dirs = _fib(4000)
    for i, fr in enumerate(fruits):
        fid = f"p{i:03d}"
        base = os.path.join(root, split, fid)
        for sub in ("input/masks", "input/poses", "input/color", "input/depth", "gt/pcd"):
            os.makedirs(os.path.join(base, sub), exist_ok=True)
        json.dump({"intrinsic_matrix": K.reshape(-1, order="F").tolist()}, open(os.path.join(base, "input/intrinsic.json"), "w"))
        sdf_obj = lambda p, f=fr: f["sdf_world"](p + f["centre"])       # fruit-centred frame
        for f in range(n_frames):
            ang = 0.25 * (f - (n_frames - 1) / 2)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            T_wc = np.eye(4)
            T_wc[:3, :3] = R
            T_wc[:3, 3] = R @ np.array([0.0, 0.0, -0.45])               # camera 0.45 m from the fruit, looking at it
            d = _render_fruit(sdf_obj, T_wc, K, img_size, np.zeros(3), 0.08)
            mask = (d > 0).astype(np.uint8)
            depth = np.where(d > 0, d, 0.9).astype(np.float32)
            name = f"{f:05d}"
            Image.fromarray(mask * 255).save(os.path.join(base, "input/masks", name + ".png"))
            np.save(os.path.join(base, "input/depth", name + ".npy"), depth)
            np.savetxt(os.path.join(base, "input/poses", name + ".txt"), T_wc)
            col = np.stack([90 + 100 * mask, 60 + 20 * mask, 40 + 0 * mask], axis=-1).astype(np.uint8)
            Image.fromarray(col).save(os.path.join(base, "input/color", name + ".png"))
        # ground truth: complete fruit surface (star-shaped level set along Fibonacci directions)
        lo, hi = np.zeros(len(dirs)), np.full(len(dirs), 0.08)
        for _ in range(20):
            mid = 0.5 * (lo + hi)
            inside = sdf_obj(dirs * mid[:, None]) < 0
            lo, hi = np.where(inside, mid, lo), np.where(inside, hi, mid)
        write_points_ply(dirs * (0.5 * (lo + hi))[:, None], os.path.join(base, "gt/pcd/fruit.ply"))
    return {"K": K, "fruits": fruits}


def _fib(n):
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)


def dump_jobs(path, named_instances, results, opt_cfg, precision):
    """Test aid of the entry-point scripts (`--dump-jobs`): the prepared per-instance inputs exactly as they went into
    `Optimizer.optimize_batch`, and its raw results, so that tests/test_gpu_cli.py can run the CPU oracle on the SAME
    inputs and compare what the scripts wrote (poses, kept / skipped instances, iteration counts, metrics)."""
    import torch
    jobs = []
    for (name, inst), res in zip(named_instances, results):
        rd = inst.render_data
        jobs.append({"name": name, "latent0": inst.latent.detach().cpu().clone(), "T_ow0": inst.T_ow.detach().cpu().clone(),
                     "points_w": inst.points_w.detach().cpu().clone(), "cube_radius": float(inst.cube_radius),
                     "pose_known": bool(inst.pose_known),
                     "render_data": None if rd is None else {k: [torch.as_tensor(a).detach().cpu().clone() for a in rd[k]]
                                                             for k in ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg")},
                     "latent": res.latent.clone(), "T_ow": res.T_ow.clone(), "iter_count": int(res.iter_count),
                     "status": int(res.status), "retried_f32": bool(getattr(res, "retried_f32", False))})
    torch.save({"jobs": jobs, "opt": opt_cfg, "precision": precision}, path)

