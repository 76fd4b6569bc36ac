"""Host-side mirror of the reference optimiser interface (`wild_completion/optimizer.py`).

`Optimizer(cfg, decoder, mesher, vis)` keeps the reference's constructor and the two entry points
`shape_pose_joint_opt` (optimizer.py:28-302) and `shape_opt_deepsdf` (:306-429) with the same argument meaning and
return triple `(latent, T_ow, iter_count)`; both are thin wrappers over the batched entry `optimize_batch`, which runs
ALL instances of a list concurrently on one GPU through `hm_optimize_batch` (libhortihip.so).  Python only packs
tensors and unpacks results; no arithmetic of the hot path happens here and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .decoder import DecoderWeights

STATUS_CONV_G, STATUS_CONV_C, STATUS_CONV_P, STATUS_MAX_ITER, STATUS_INVALID, STATUS_SOLVE_FAILED = 1, 2, 4, 8, 16, 32
STATUS_FRAME_SKIPPED = 64       # informational: a frame was left out in some iteration (optimizer.py:130-132)

_vp = ctypes.c_void_p


class HmOptCfg(ctypes.Structure):
    _fields_ = [("scale_on", ctypes.c_int), ("robust_iter", ctypes.c_int), ("lm_on", ctypes.c_int),
                ("lm_eye", ctypes.c_int), ("lm_lambda_0", ctypes.c_float), ("s_damp", ctypes.c_float),
                ("recon_robust_th", ctypes.c_float), ("render_robust_th", ctypes.c_float),
                ("n_sample_on_ray", ctypes.c_int), ("log_sdf_occ", ctypes.c_int), ("occ_cutoff", ctypes.c_float),
                ("occlusion_on", ctypes.c_int), ("w_recon", ctypes.c_float), ("w_depth", ctypes.c_float),
                ("w_mask", ctypes.c_float), ("w_codereg", ctypes.c_float), ("max_iter", ctypes.c_int),
                ("epsilon_g", ctypes.c_float), ("epsilon_c", ctypes.c_float), ("epsilon_t", ctypes.c_float),
                ("epsilon_r", ctypes.c_float), ("epsilon_s", ctypes.c_float), ("occlusion_th", ctypes.c_float),
                ("min_valid_sample", ctypes.c_int), ("min_grad_thre", ctypes.c_float)]


class HmLimits(ctypes.Structure):
    _fields_ = [("max_batch", ctypes.c_int), ("max_points", ctypes.c_int), ("max_frames", ctypes.c_int),
                ("max_rays", ctypes.c_int), ("max_samples", ctypes.c_int), ("max_grad_samples", ctypes.c_int)]


class HmBatch(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("points_stride", ctypes.c_int), ("d_points_w", _vp), ("d_n_points", _vp),
                ("d_T_wc", _vp), ("d_rays", _vp), ("d_depth", _vp), ("d_n_fg", _vp), ("d_n_bg", _vp),
                ("d_n_frames", _vp), ("d_cube_radius", _vp), ("d_pose_known", _vp), ("d_latent", _vp),
                ("d_T_ow", _vp), ("d_iter_count", _vp), ("d_status", _vp)]


class HmDebug(ctypes.Structure):
    _fields_ = [("d_A", _vp), ("d_b", _vp), ("d_delta", _vp), ("d_counts", _vp)]


def _declare_opt(lib):
    if getattr(lib, "_hm_opt_declared", False):
        return
    lib.hm_workspace_create.restype = ctypes.c_int
    lib.hm_workspace_create.argtypes = [_vp, ctypes.POINTER(HmLimits), ctypes.POINTER(_vp)]
    lib.hm_workspace_destroy.restype = ctypes.c_int
    lib.hm_workspace_destroy.argtypes = [_vp]
    lib.hm_workspace_bytes.restype = ctypes.c_size_t
    lib.hm_workspace_bytes.argtypes = [_vp]
    lib.hm_optimize_batch.restype = ctypes.c_int
    lib.hm_optimize_batch.argtypes = [_vp, ctypes.POINTER(HmOptCfg), ctypes.POINTER(HmBatch), ctypes.c_int,
                                      ctypes.POINTER(HmDebug), _vp]
    lib.hm_render_residuals.restype = ctypes.c_int
    lib.hm_render_residuals.argtypes = [_vp, ctypes.POINTER(HmOptCfg), ctypes.POINTER(HmBatch), _vp, _vp, _vp, _vp, _vp, _vp]
    lib._hm_opt_declared = True


def opt_cfg_from_dict(opt: dict) -> HmOptCfg:
    """`opt`: the reference's `cfg['opt']` block (configs/*.yaml); string-y floats are cast like optimizer.py:32-52."""
    c = HmOptCfg()
    c.scale_on = int(bool(opt["scale_on"]))
    c.robust_iter = int(opt["robust_iter"])
    c.lm_on = int(bool(opt["lm"]["lm_on"]))
    c.lm_eye = int(bool(opt["lm"]["lm_eye"]))
    c.lm_lambda_0 = float(opt["lm"]["lm_lambda_0"])
    c.s_damp = float(opt["lm"]["s_damp"])
    c.recon_robust_th = float(opt["recon"]["robust_th_m"])
    c.render_robust_th = float(opt["render"]["robust_th_m"])
    c.n_sample_on_ray = int(opt["render"]["n_sample_on_ray"])
    c.log_sdf_occ = int(bool(opt["render"]["log_sdf_occ"]))
    c.occ_cutoff = float(opt["render"]["occ_cutoff_m"])
    c.occlusion_on = int(bool(opt["render"]["occlusion_on"]))
    c.w_recon = float(opt["weight"]["w_recon"])
    c.w_depth = float(opt["weight"]["w_depth"])
    c.w_mask = float(opt["weight"]["w_mask"])
    c.w_codereg = float(opt["weight"]["w_codereg"])
    cv = opt["converge"]
    c.max_iter = int(cv["max_iter"])
    c.epsilon_g = float(cv["epsilon_g"])
    c.epsilon_c = float(cv["epsilon_c"])
    c.epsilon_t = float(cv.get("epsilon_t", 0.0))
    c.epsilon_r = float(cv.get("epsilon_r", 0.0))
    c.epsilon_s = float(cv.get("epsilon_s", 0.0))
    c.occlusion_th = 0.03        # loss.py:11
    c.min_valid_sample = 100     # loss.py:11
    c.min_grad_thre = 1e-6       # loss.py:11
    return c


@dataclass
class Instance:
    """One fruit instance, in the reference's caller-side format (test_wild_completion.py:154-226)."""
    latent: torch.Tensor                 # (L,)
    T_ow: torch.Tensor                   # (4,4)
    points_w: torch.Tensor               # (N,3)
    render_data: Optional[dict] = None   # keys T_wc, rays_fg, rays_bg, depth_fg, depth_bg: lists per frame
    cube_radius: float = 0.08
    pose_known: bool = False


@dataclass
class Result:
    latent: torch.Tensor
    T_ow: torch.Tensor
    iter_count: int
    status: int


def select_frames(n_all: int, n_frame: int) -> np.ndarray:
    """optimizer.py:77-78: np.linspace(0, F_all-1, min(n_frame, F_all)).astype(int32)."""
    return np.linspace(0, n_all - 1, min(int(n_frame), n_all)).astype(np.int32)


class Workspace:
    def __init__(self, dec: DecoderWeights, max_batch, max_points, max_frames=0, max_rays=0, max_samples=0,
                 max_grad_samples=0):
        lib = _lib.lib()
        _declare_opt(lib)
        self.dec = dec
        self.limits = HmLimits(int(max_batch), int(max_points), int(max_frames), int(max_rays), int(max_samples),
                               int(max_grad_samples))
        h = _vp()
        _lib.check(lib.hm_workspace_create(dec.handle, ctypes.byref(self.limits), ctypes.byref(h)),
                   "hm_workspace_create")
        self.handle = h

    def fits(self, B, n_pts, F, R, M):
        l = self.limits
        return (B <= l.max_batch and n_pts <= l.max_points and F <= max(l.max_frames, 0) and R <= max(l.max_rays, 0)
                and M <= max(l.max_samples, 0))

    @property
    def nbytes(self):
        return int(_lib.lib().hm_workspace_bytes(self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().hm_workspace_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class PackedBatch:
    """Padded device tensors for a list of instances (the layout `hm_batch` documents)."""

    def __init__(self, instances: Sequence[Instance], L: int, n_frame: int, device, F_cap=None, R_cap=None,
                 N_cap=None, joint=True):
        B = len(instances)
        self.B = B
        f32, i32 = torch.float32, torch.int32
        n_pts = [int(inst.points_w.shape[0]) for inst in instances]
        N = max(N_cap or 0, max(n_pts))
        pts = torch.zeros(B, N, 3, dtype=f32)
        for b, inst in enumerate(instances):
            pts[b, :n_pts[b]] = inst.points_w.detach().to("cpu", f32)
        self.points_stride = N
        self.points_w = pts.to(device)
        self.n_points = torch.tensor(n_pts, dtype=i32, device=device)
        self.latent = torch.stack([inst.latent.detach().to("cpu", f32).reshape(L) for inst in instances]).contiguous().to(device)
        self.T_ow = torch.stack([inst.T_ow.detach().to("cpu", f32).reshape(16) for inst in instances]).contiguous().to(device)
        self.cube_radius = torch.tensor([float(inst.cube_radius) for inst in instances], dtype=f32, device=device)
        self.pose_known = torch.tensor([int(bool(inst.pose_known)) for inst in instances], dtype=i32, device=device)
        self.iter_count = torch.zeros(B, dtype=i32, device=device)
        self.status = torch.zeros(B, dtype=i32, device=device)
        self.F = self.R = 0
        self.T_wc = self.rays = self.depth = self.n_fg = self.n_bg = self.n_frames = None
        if joint:
            sel = []
            for inst in instances:
                rd = inst.render_data
                sel.append(select_frames(len(rd["T_wc"]), n_frame))
            F = max(F_cap or 0, max(len(s) for s in sel), 1)
            R = max(R_cap or 0, 1)
            for inst, s in zip(instances, sel):
                for idx in s:
                    R = max(R, int(inst.render_data["rays_fg"][idx].shape[0] + inst.render_data["rays_bg"][idx].shape[0]))
            T_wc = torch.zeros(B, F, 16, dtype=f32)
            rays = torch.zeros(B, F, R, 3, dtype=f32)
            depth = torch.zeros(B, F, R, dtype=f32)
            n_fg = torch.zeros(B, F, dtype=i32)
            n_bg = torch.zeros(B, F, dtype=i32)
            n_frames = torch.zeros(B, dtype=i32)
            for b, (inst, s) in enumerate(zip(instances, sel)):
                rd = inst.render_data
                n_frames[b] = len(s)
                for k, idx in enumerate(s):
                    fg, bg = rd["rays_fg"][idx].detach().to("cpu", f32), rd["rays_bg"][idx].detach().to("cpu", f32)
                    nf, nb = fg.shape[0], bg.shape[0]
                    T_wc[b, k] = rd["T_wc"][idx].detach().to("cpu", f32).reshape(16)
                    rays[b, k, :nf] = fg
                    rays[b, k, nf:nf + nb] = bg
                    depth[b, k, :nf] = rd["depth_fg"][idx].detach().to("cpu", f32)
                    depth[b, k, nf:nf + nb] = rd["depth_bg"][idx].detach().to("cpu", f32)
                    n_fg[b, k] = nf
                    n_bg[b, k] = nb
            self.F, self.R = F, R
            self.T_wc, self.rays, self.depth = T_wc.to(device), rays.to(device), depth.to(device)
            self.n_fg, self.n_bg, self.n_frames = n_fg.to(device), n_bg.to(device), n_frames.to(device)

    def as_struct(self) -> HmBatch:
        p = lambda t: 0 if t is None else t.data_ptr()
        return HmBatch(self.B, self.points_stride, p(self.points_w), p(self.n_points), p(self.T_wc), p(self.rays),
                       p(self.depth), p(self.n_fg), p(self.n_bg), p(self.n_frames), p(self.cube_radius),
                       p(self.pose_known), p(self.latent), p(self.T_ow), p(self.iter_count), p(self.status))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def run_packed(ws: Workspace, cfg: HmOptCfg, pb: PackedBatch, mode: int, debug: Optional[dict] = None):
    """Enqueue the whole optimisation of a packed batch (asynchronous w.r.t. the host)."""
    lib = _lib.lib()
    _declare_opt(lib)
    bs = pb.as_struct()
    dbg = None
    if debug is not None:
        L = ws.dec.latent_dim
        dev = pb.latent.device
        debug["A"] = torch.zeros(pb.B, L + 8, L + 8, device=dev)
        debug["b"] = torch.zeros(pb.B, L + 8, device=dev)
        debug["delta"] = torch.zeros(pb.B, L + 8, device=dev)
        debug["counts"] = torch.zeros(pb.B, 4, dtype=torch.int32, device=dev)
        dbg = HmDebug(debug["A"].data_ptr(), debug["b"].data_ptr(), debug["delta"].data_ptr(),
                      debug["counts"].data_ptr())
    rc = lib.hm_optimize_batch(ws.handle, ctypes.byref(cfg), ctypes.byref(bs), mode,
                               ctypes.byref(dbg) if dbg is not None else None, _stream())
    _lib.check(rc, "hm_optimize_batch")


def optimize_batch(dec: DecoderWeights, opt: dict, instances: Sequence[Instance], shape_only: bool = False,
                   workspace: Optional[Workspace] = None, device="cuda", debug: Optional[dict] = None) -> List[Result]:
    """Optimise all `instances` concurrently; results are returned in input order (identical instance indexing)."""
    if len(instances) == 0:
        return []
    cfg = opt_cfg_from_dict(opt)
    L = dec.latent_dim
    pb = PackedBatch(instances, L, int(opt["render"]["n_frame"]), device, joint=not shape_only)
    M = cfg.n_sample_on_ray
    if workspace is None or not workspace.fits(pb.B, pb.points_stride, pb.F, pb.R, M if not shape_only else 0):
        workspace = Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, 0 if shape_only else M)
    elif not shape_only:
        # the workspace strides are capacities: re-pack to the workspace's frame/ray capacity
        l = workspace.limits
        if (pb.F, pb.R) != (l.max_frames, l.max_rays):
            pb = PackedBatch(instances, L, int(opt["render"]["n_frame"]), device, F_cap=l.max_frames, R_cap=l.max_rays)
    run_packed(workspace, cfg, pb, 1 if shape_only else 0, debug)
    lat, T, it, st = pb.latent.cpu(), pb.T_ow.cpu(), pb.iter_count.cpu(), pb.status.cpu()
    return [Result(lat[b].clone(), T[b].reshape(4, 4).clone(), int(it[b]), int(st[b])) for b in range(pb.B)]


def optimize_grouped(jobs: Sequence[tuple], shape_only: bool = False, device="cuda") -> List[Result]:
    """Mixed workloads (BASELINE.json configs[4]: pepper + berry decoders, different YAML blocks in one job list).
    `jobs` is a list of (DecoderWeights, opt_dict, Instance).  Instances are grouped by (decoder, config) so that each
    batch runs with ONE resident weight set and ONE option block; groups run back to back and the results are
    scattered back so that result i belongs to job i (identical instance indexing)."""
    groups = {}
    for i, (dec, opt, inst) in enumerate(jobs):
        groups.setdefault((id(dec), id(opt)), (dec, opt, []))[2].append((i, inst))
    out: List[Optional[Result]] = [None] * len(jobs)
    for dec, opt, members in groups.values():
        res = optimize_batch(dec, opt, [m[1] for m in members], shape_only, None, device)
        for (i, _), r in zip(members, res):
            out[i] = r
    return out


class Optimizer(object):
    """Drop-in for `wild_completion.optimizer.Optimizer` (optimizer.py:16-25)."""

    def __init__(self, cfg, decoder, mesher=None, vis=None):
        self.dev = cfg.get("device", "cuda")
        self.dtype = torch.float32
        self.opt_cfg = cfg["opt"]
        if isinstance(decoder, DecoderWeights):
            self.decoder = decoder
        else:                                   # the reference passes an nn.Module (optimizer.py:17)
            self.decoder = DecoderWeights.from_module(decoder)
        self.mesher = mesher
        self.vis = vis
        self.log_on = cfg.get("vis", {}).get("log_on", False)
        self._ws = None

    def _device(self):
        return "cuda" if str(self.dev).startswith("cuda") else self.dev

    def optimize_batch(self, instances: Sequence[Instance], shape_only: bool = False) -> List[Result]:
        return optimize_batch(self.decoder, self.opt_cfg, instances, shape_only, None, self._device())

    def shape_pose_joint_opt(self, latent, T_ow_torch, render_data, points_w_torch, cube_radius, cur_color=None,
                             pose_known=False):
        """optimizer.py:28-302.  `latent` is updated in place AND returned, like the reference (:248,302)."""
        inst = Instance(latent, T_ow_torch, points_w_torch, render_data, float(cube_radius), bool(pose_known))
        res = self.optimize_batch([inst], shape_only=False)[0]
        if self.log_on:                                   # the reference's console messages (:131, :140)
            if res.status & STATUS_FRAME_SKIPPED:
                print("This frame is not valid")
            if res.status & STATUS_INVALID:
                print("This submap is not valid")
        latent.data.copy_(res.latent.to(latent.device, latent.dtype))
        return latent, res.T_ow.to(T_ow_torch.device, T_ow_torch.dtype), res.iter_count

    def shape_opt_deepsdf(self, latent, T_ow_torch, points_w_torch, cur_color=None):
        """optimizer.py:306-429 (pose frozen, SDF term + code regulariser)."""
        inst = Instance(latent, T_ow_torch, points_w_torch, None, 0.08, True)
        res = self.optimize_batch([inst], shape_only=True)[0]
        latent.data.copy_(res.latent.to(latent.device, latent.dtype))
        return latent, T_ow_torch, res.iter_count
