"""Functional mirrors of `wild_completion/loss.py` (residuals + analytic Jacobians), GPU-backed, same signatures.

`compute_sdf_loss` -> hm_decode_batch (mode 1); `compute_render_loss` -> hm_render_residuals with the caller's
`t_obj_cam`, `sampled_ray_depth` window and `object_bbx_radius` passed through the frame override."""
from __future__ import annotations

import ctypes

import torch

from . import _lib, ops
from .optimizer import HmBatch, HmOptCfg, Workspace, _declare_opt
from .utils import _pack_points, as_weights


def compute_sdf_loss(decoder, latent_vector, pts_surface_obj, scale_on=False):
    """`loss.py:219-243` -> (res (N,1,1), jac_recon_tow (N,1,6|7), jac_recon_code (N,1,C))."""
    dec = as_weights(decoder)
    dev = torch.device("cuda")
    L = dec.latent_dim
    P = 7 if scale_on else 6
    pts4, n = _pack_points(pts_surface_obj, dev)
    lat = latent_vector.detach().to(dev, torch.float32).reshape(1, -1).contiguous()
    y, J = ops.decode_batch(dec, lat, pts4, torch.tensor([n], dtype=torch.int32, device=dev), mode=1, pose_dim=P)
    return (y[0, :n].reshape(n, 1, 1), J[0, :n, L:L + P].reshape(n, 1, P).contiguous(),
            J[0, :n, :L].reshape(n, 1, L).contiguous())


def compute_render_loss(decoder, latent_vector, ray_directions, depth_obs_fg, depth_obs_bg, t_obj_cam,
                        sampled_ray_depth, scale_on=False, log_occ_on=False, occupancy_th=0.01,
                        object_bbx_radius=0.1, occlusion_on=True, occlusion_th=0.03, min_valid_sample=100,
                        min_grad_thre=1e-6):
    """`loss.py:8-217`.  Returns None (too few ball-valid samples, :43-45) or the reference's six tensors
    (res_d (V,1,1), jac_d_tow (V,1,P), jac_d_code (V,1,C), res_m, jac_m_tow, jac_m_code), rays in ascending index.
    `sampled_ray_depth` must be a linspace (it always is: optimizer.py:111); only its end points are used."""
    dec = as_weights(decoder)
    dev = torch.device("cuda")
    lib = _lib.lib()
    _declare_opt(lib)
    L = dec.latent_dim
    P = 7 if scale_on else 6
    f32, i32 = torch.float32, torch.int32
    rays = ray_directions.detach().to(dev, f32).reshape(-1, 3)
    R = rays.shape[0]
    n_fg, n_bg = int(depth_obs_fg.shape[0]), int(depth_obs_bg.shape[0])
    assert n_fg + n_bg == R
    M = int(sampled_ray_depth.shape[0])
    ws = Workspace(dec, 1, 64, 1, R, M)
    cfg = HmOptCfg()
    cfg.scale_on = int(scale_on); cfg.n_sample_on_ray = M; cfg.log_sdf_occ = int(log_occ_on)
    cfg.occ_cutoff = float(occupancy_th); cfg.occlusion_on = int(occlusion_on)
    cfg.occlusion_th = float(occlusion_th); cfg.min_valid_sample = int(min_valid_sample)
    cfg.min_grad_thre = float(min_grad_thre); cfg.max_iter = 1
    depth = torch.cat([depth_obs_fg.detach().to(dev, f32).reshape(-1), depth_obs_bg.detach().to(dev, f32).reshape(-1)])
    T = t_obj_cam.detach().to(dev, f32)
    frame = torch.zeros(1, 1, 16, device=dev, dtype=f32)
    frame[0, 0, :12] = T[:3, :].reshape(12)
    frame[0, 0, 12] = float(sampled_ray_depth[0])
    frame[0, 0, 13] = float(sampled_ray_depth[-1])
    frame[0, 0, 14] = float(object_bbx_radius)
    lat = latent_vector.detach().to(dev, f32).reshape(1, L).contiguous()
    dummy_pts = torch.zeros(1, 64, 3, device=dev, dtype=f32)
    n_pts = torch.zeros(1, dtype=i32, device=dev)
    T_ow = torch.eye(4, device=dev, dtype=f32).reshape(1, 16).contiguous()
    T_wc = torch.eye(4, device=dev, dtype=f32).reshape(1, 1, 16).contiguous()
    nfg = torch.tensor([[n_fg]], dtype=i32, device=dev)
    nbg = torch.tensor([[n_bg]], dtype=i32, device=dev)
    nfr = torch.tensor([1], dtype=i32, device=dev)
    cube = torch.tensor([float(object_bbx_radius)], dtype=f32, device=dev)
    itc = torch.zeros(1, dtype=i32, device=dev)
    stt = torch.zeros(1, dtype=i32, device=dev)
    rays_b = rays.reshape(1, 1, R, 3).contiguous()
    depth_b = depth.reshape(1, 1, R).contiguous()
    bs = HmBatch(1, 64, dummy_pts.data_ptr(), n_pts.data_ptr(), T_wc.data_ptr(), rays_b.data_ptr(), depth_b.data_ptr(),
                 nfg.data_ptr(), nbg.data_ptr(), nfr.data_ptr(), cube.data_ptr(), 0, lat.data_ptr(), T_ow.data_ptr(),
                 itc.data_ptr(), stt.data_ptr())
    ldJ = L + 8
    rows = torch.zeros(1, 2 * R, ldJ, device=dev, dtype=f32)
    V = torch.zeros(1, dtype=i32, device=dev)
    counts = torch.zeros(1, 4, dtype=i32, device=dev)
    rc = lib.hm_render_residuals(ws.handle, ctypes.byref(cfg), ctypes.byref(bs), frame.data_ptr(), rows.data_ptr(),
                                 V.data_ptr(), 0, counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "hm_render_residuals")
    cnt = counts.cpu()
    if int(cnt[0, 0]) < int(min_valid_sample):
        return None
    v = int(V.item())
    d, m = rows[0, :v], rows[0, R:R + v]
    return (d[:, L + 7].reshape(v, 1, 1).clone(), d[:, L:L + P].reshape(v, 1, P).clone(), d[:, :L].reshape(v, 1, L).clone(),
            m[:, L + 7].reshape(v, 1, 1).clone(), m[:, L:L + P].reshape(v, 1, P).clone(), m[:, :L].reshape(v, 1, L).clone())
