"""Synthetic fruit decoders and synthetic fruit instances (numpy only).

The shipped reference tree carries no decoder weights and no dataset
(`/root/reference/.MISSING_LARGE_BLOBS`), and a random-weight DeepSDF decoder has
no zero level set, so the optimisation loop degenerates at iteration 0
(reference `wild_completion/optimizer.py:139-141`).  Benchmarks, parity tests and
golden vectors therefore use an *analytic* decoder with the reference architecture
(`deepsdf/networks/deep_sdf_decoder.py:10-72`, `deepsdf/models/sweetpepper_32/specs.json`):

    sdf(x, z) ~= 4 * mean_k relu(u_k . x + B(u_k) . z) - r0      ("lumpy sphere")

realised with the reference's 9 Linear layers (8 hidden of width 512, skip concat
at layer 4) so that it exercises exactly the same arithmetic as a trained model.
All draws come from ``np.random.RandomState(seed)`` in a fixed order, so the same
weights are regenerated bit-identically on the build box and on the GPU box.

This module is a *data generator*; it is not the oracle and not the hot path.
"""
from __future__ import annotations

import numpy as np

HIDDEN = 512
N_LIN = 9           # lin0..lin8
SKIP_LAYER = 4      # latent_in = [4]


def layer_shapes(latent_dim: int, hidden: int = HIDDEN):
    """(out, in) of lin0..lin8 for the reference architecture (specs.json:7-18)."""
    d0 = latent_dim + 3
    m = hidden - d0
    if m <= 0:
        raise ValueError("latent_dim + 3 must be < hidden width")
    shp = [(hidden, d0), (hidden, hidden), (hidden, hidden), (m, hidden)]
    shp += [(hidden, hidden)] * 4
    shp += [(1, hidden)]
    return shp


def make_synthetic_decoder(latent_dim: int, seed: int = 0, r0: float = 0.04,
                           aniso=(1.0, 1.0, 1.0), latent_gain: float = 0.1,
                           noise: float = 0.01, hidden: int = HIDDEN,
                           wn_perturb: float = 0.0, freq_sigma: float = 2.0, bias_sigma: float = 0.0):
    """Return a dict of fp32 arrays describing a weight-normed decoder.

    Keys: ``lin{l}.weight_v``, ``lin{l}.weight_g`` (l = 0..7), ``lin8.weight``,
    ``lin{l}.bias`` (l = 0..8), plus ``latent_dim``/``hidden``.  The effective
    weight of a weight-normed layer is ``g * v / ||v||_row`` (reference
    `deep_sdf_decoder.py:49-54`, torch ``weight_norm`` dim=0).  With
    ``wn_perturb == 0`` we set ``g = ||v||`` so the effective weight is ``v``;
    a non-zero value perturbs ``g`` to exercise the folding code.  ``bias_sigma`` adds Gaussian noise to the hidden
    biases from a SEPARATE stream (seed + 7919), so the weight draws of existing fixtures do not move.
    """
    L, H = int(latent_dim), int(hidden)
    rs = np.random.RandomState(seed)
    shp = layer_shapes(L, H)
    m = shp[3][0]
    an = np.asarray(aniso, dtype=np.float64)

    def unit_dirs(n):
        u = rs.randn(n, 3)
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        return u

    # The latent acts through smooth random-Fourier features of the half-space direction,
    # B[k, j] = gain/sqrt(L) * sqrt(2) cos(omega_j . u_k + psi_j): z then deforms the support
    # function of the shape coherently (d sdf / d z_j = O(gain/sqrt(L))), like a trained
    # DeepSDF latent does, instead of cancelling out over directions.
    omega = freq_sigma * rs.randn(L, 3)
    psi = rs.uniform(0, 2 * np.pi, L)

    def latent_rows(u):
        return latent_gain / np.sqrt(L) * np.sqrt(2.0) * np.cos(u @ omega.T + psi)

    W = {}
    # lin0: rows [B_k (L) | u_k (3)]
    U0 = unit_dirs(H)
    # anisotropic scaling of the direction rows makes rotation observable (SURVEY 8d note 3)
    W[0] = np.concatenate([latent_rows(U0), U0 * an], axis=1)
    for l in (1, 2):
        W[l] = np.eye(H) + noise * rs.randn(H, H) / np.sqrt(H)
    W[3] = np.eye(H)[:m] + noise * rs.randn(m, H) / np.sqrt(H)
    # lin4: input [h3 (m) | z (L) | x (3)] -> [h3 passthrough (m) ; new directions (L+3)]
    W4 = np.zeros((H, H))
    W4[:m, :m] = np.eye(m)
    U4 = unit_dirs(H - m)
    W4[m:, m:m + L] = latent_rows(U4)
    W4[m:, m + L:] = U4 * an
    W4[:, :m] += noise * rs.randn(H, m) / np.sqrt(H)
    W[4] = W4
    for l in (5, 6, 7):
        W[l] = np.eye(H) + noise * rs.randn(H, H) / np.sqrt(H)
    W[8] = np.full((1, H), 4.0 / H)

    out = {"latent_dim": L, "hidden": H}
    rsb = np.random.RandomState(seed + 7919)
    for l in range(N_LIN):
        w = W[l].astype(np.float32)
        b = np.zeros(w.shape[0], dtype=np.float32)
        if bias_sigma and l < 8:
            b = (bias_sigma * rsb.randn(w.shape[0])).astype(np.float32)
        if l == 8:
            b[0] = -r0
            out["lin8.weight"] = w
        else:
            g = np.linalg.norm(w.astype(np.float64), axis=1, keepdims=True)
            if wn_perturb:
                g = g * (1.0 + wn_perturb * rs.randn(*g.shape))
            out[f"lin{l}.weight_v"] = w
            out[f"lin{l}.weight_g"] = g.astype(np.float32)
        out[f"lin{l}.bias"] = b
    return out


def fold_weight_norm(params):
    """Effective (W_l, b_l), l = 0..8, fp32: ``W = g * v / ||v||_row`` folded once.

    Mirrors what torch's weight_norm pre-hook recomputes on every forward
    (reference `deep_sdf_decoder.py:49-54`); the norm is taken in fp32 like torch.
    """
    Ws, bs = [], []
    for l in range(N_LIN):
        if f"lin{l}.weight_v" in params:
            v = np.asarray(params[f"lin{l}.weight_v"], dtype=np.float32)
            g = np.asarray(params[f"lin{l}.weight_g"], dtype=np.float32).reshape(-1, 1)
            nrm = np.sqrt((v.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))
            w = (v * (g / nrm)).astype(np.float32)
        else:
            w = np.asarray(params[f"lin{l}.weight"], dtype=np.float32)
        Ws.append(np.ascontiguousarray(w))
        bs.append(np.ascontiguousarray(np.asarray(params[f"lin{l}.bias"], dtype=np.float32)))
    return Ws, bs


def np_decoder_forward(Ws, bs, z, x, dtype=np.float64):
    """Plain numpy forward of the folded decoder, used only to *generate* scenes."""
    z = np.asarray(z, dtype=dtype)
    x = np.asarray(x, dtype=dtype).reshape(-1, 3)
    u = np.concatenate([np.broadcast_to(z, (x.shape[0], z.shape[0])), x], axis=1)
    h = u
    for l in range(N_LIN):
        if l == SKIP_LAYER:
            h = np.concatenate([h, u], axis=1)
        h = h @ Ws[l].astype(dtype).T + bs[l].astype(dtype)
        if l < N_LIN - 1:
            h = np.maximum(h, 0)
    return np.tanh(h[:, 0])


CAM_K = np.array([[600.0, 0.0, 320.0], [0.0, 600.0, 240.0], [0.0, 0.0, 1.0]])


def _first_hit(sdf_fn, origins, dirs, t0, t1, n_march=48, n_bisect=18):
    """March + bisect for the first sign change of sdf along origin + t*dir."""
    n = origins.shape[0]
    ts = np.linspace(t0, t1, n_march)
    vals = np.stack([sdf_fn(origins + dirs * t) for t in ts], axis=1)  # (n, n_march)
    neg = vals < 0
    hit = neg.any(axis=1)
    first = np.argmax(neg, axis=1)
    first = np.maximum(first, 1)
    lo = ts[first - 1]
    hi = ts[first]
    for _ in range(n_bisect):
        mid = 0.5 * (lo + hi)
        v = sdf_fn(origins + dirs * mid[:, None])
        inside = v < 0
        hi = np.where(inside, mid, hi)
        lo = np.where(inside, lo, mid)
    t = 0.5 * (lo + hi)
    hit &= ~neg[:, 0]
    return hit, t


def make_instance(Ws, bs, latent_dim, inst_id, n_pts=2048, n_frames=1, n_fg=32, n_bg=32,
                  seed_base=1000, r_max=0.08, sdf_fn_factory=None, z_sigma=0.07,
                  pose_noise=0.005, pix_halfwidth=80.0, scale_init=1.0, z_true=None, baseline=0.02):
    """One synthetic fruit instance in the reference's caller-side data format.

    Returns a dict with fp32 arrays: ``latent0 (L,)``, ``T_ow0 (4,4)``, ``points_w (n_pts,3)``,
    and ``render`` = {"T_wc": [F x (4,4)], "rays_fg": [...(n_fg,3)], "rays_bg": [...],
    "depth_fg": [...], "depth_bg": [...]} -- the keys the reference optimiser consumes
    (`wild_completion/optimizer.py:77-83`, produced by `utils.py:96-105`).  Also the
    generating truth (``z_true``, ``T_wo_true``) for Chamfer/pose-error metrics.

    ``sdf_fn_factory(z)`` may supply an accelerated ``x -> sdf`` callable; default is the
    numpy forward above.  ``z_true`` overrides the Gaussian draw of the generating latent (used with trained
    decoders, whose plausible shapes sit near their learnt codes); the draw is still consumed so that the
    remaining stream does not move.
    """
    L = int(latent_dim)
    rs = np.random.RandomState(seed_base + int(inst_id))
    z_draw = (z_sigma * rs.randn(L)).astype(np.float32)
    z_true = z_draw if z_true is None else np.asarray(z_true, dtype=np.float32).reshape(L)
    centre = rs.uniform(-0.01, 0.01, 3) + np.array([0.0, 0.0, 0.5])
    if sdf_fn_factory is None:
        def sdf_obj(p):
            return np_decoder_forward(Ws, bs, z_true, p)
    else:
        sdf_obj = sdf_fn_factory(z_true)

    def sdf_world(p):
        return sdf_obj(p - centre)

    # surface points on the camera-facing cap (camera at world origin looking +z)
    pts = []
    need = n_pts
    while need > 0:
        d = rs.randn(2 * need + 16, 3)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = d[d[:, 2] < 0.2][:need]
        # march from outside (r_max) towards the centre
        org = centre + d * r_max
        hit, t = _first_hit(sdf_world, org, -d, 0.0, r_max)
        p = org - d * t[:, None]
        pts.append(p[hit])
        need -= int(hit.sum())
    points_w = np.concatenate(pts, axis=0)[:n_pts]
    points_w = points_w + 1e-3 * rs.randn(*points_w.shape)

    Kinv = np.linalg.inv(CAM_K)
    render = {"T_wc": [], "rays_fg": [], "rays_bg": [], "depth_fg": [], "depth_bg": []}
    for f in range(n_frames):
        T_wc = np.eye(4)
        if f > 0:
            T_wc[:3, 3] = np.array([baseline * f, -0.5 * baseline * f, 0.0])     # camera offsets (default 2 cm, -1 cm per frame)
        cam_o = T_wc[:3, 3]
        pc = centre - cam_o
        uv_c = (CAM_K @ (pc / pc[2]))[:2]
        fg_r, fg_d, bg_r, bg_d = [], [], [], []
        guard = 0
        while (len(fg_r) < n_fg or len(bg_r) < n_bg) and guard < 64:
            guard += 1
            nb = 4 * (n_fg + n_bg)
            uv = uv_c + rs.uniform(-pix_halfwidth, pix_halfwidth, (nb, 2))
            dirs = (np.concatenate([uv, np.ones((nb, 1))], axis=1)[:, None, :] * Kinv).sum(-1)
            hit, t = _first_hit(sdf_world, np.broadcast_to(cam_o, dirs.shape).copy(), dirs,
                                pc[2] - r_max, pc[2] + r_max)
            for i in range(nb):
                if hit[i] and len(fg_r) < n_fg:
                    fg_r.append(dirs[i]); fg_d.append(t[i])
                elif (not hit[i]) and len(bg_r) < n_bg:
                    bg_r.append(dirs[i])
                    bg_d.append(0.3 if rs.rand() < 0.15 else 0.9)  # some occluders in front
        render["T_wc"].append(T_wc.astype(np.float32))
        render["rays_fg"].append(np.asarray(fg_r, dtype=np.float32).reshape(-1, 3))
        render["rays_bg"].append(np.asarray(bg_r, dtype=np.float32).reshape(-1, 3))
        render["depth_fg"].append(np.asarray(fg_d, dtype=np.float32))
        render["depth_bg"].append(np.asarray(bg_d, dtype=np.float32))

    T_wo_true = np.eye(4)
    T_wo_true[:3, 3] = centre
    T_wo0 = np.eye(4)
    T_wo0[:3, :3] *= scale_init
    T_wo0[:3, 3] = centre + pose_noise * rs.randn(3)
    T_ow0 = np.linalg.inv(T_wo0)
    return {
        "id": int(inst_id),
        "latent0": np.zeros(L, dtype=np.float32),
        "T_ow0": T_ow0.astype(np.float32),
        "points_w": points_w.astype(np.float32),
        "render": render,
        "cube_radius": float(r_max),
        "z_true": z_true,
        "T_wo_true": T_wo_true.astype(np.float32),
    }


# --------------------------------------------------------------------------------------
# decoders of OTHER layer tables (any `Decoder(latent_size, dims, ...)` of deep_sdf_decoder.py:11-72)
# --------------------------------------------------------------------------------------
def arch_layer_dims(latent_dim, dims, latent_in=(), xyz_in_all=False):
    """[(out, in)] of lin0..lin{n-1} as `Decoder.__init__` sizes them (deep_sdf_decoder.py:29-47)."""
    full = [latent_dim + 3] + list(dims) + [1]
    n = len(full)
    shp = []
    for layer in range(n - 1):
        if layer + 1 in latent_in:
            od = full[layer + 1] - full[0]
        else:
            od = full[layer + 1]
            if xyz_in_all and layer != n - 2:
                od -= 3
        shp.append((od, full[layer]))
    return shp


def make_arch_decoder(latent_dim, dims, latent_in=(), norm_layers=(), weight_norm=False, xyz_in_all=False,
                      use_tanh=False, seed=0, analytic=False, r0=0.04, aniso=(1.0, 0.75, 1.3), latent_gain=0.1,
                      noise=0.01, freq_sigma=2.0):
    """Parameters, in the reference's state-dict layout, of `Decoder(latent_dim, dims, norm_layers=..., latent_in=...,
    weight_norm=..., xyz_in_all=..., use_tanh=...)`: `lin{l}.weight_v/_g` for weight-normed layers (`weight_norm` and l in
    `norm_layers`, deep_sdf_decoder.py:49-54), `lin{l}.weight` otherwise, `lin{l}.bias`, and `bn{l}.weight/.bias` for the
    LayerNorm modules of `norm_layers` without `weight_norm` (:57-62).  Plus 'latent_dim' and 'use_tanh'.

    analytic=False: He-scaled random weights (the function is arbitrary; used for decode / Jacobian parity).
    analytic=True (no LayerNorm): the "lumpy sphere" of `make_synthetic_decoder` realised in this layer table -- half-space
    features in lin0, every later hidden layer passes the running mean on, `4 * mean - r0` at the end -- so that the LM
    loop has a fruit to fit.  All draws come from np.random.RandomState(seed) in a fixed order."""
    L = int(latent_dim)
    rs = np.random.RandomState(seed)
    shp = arch_layer_dims(L, dims, latent_in, xyz_in_all)
    n = len(shp)
    has_ln = [(not weight_norm) and (l in norm_layers) and l < n - 1 for l in range(n)]
    if analytic and any(has_ln):
        raise ValueError("the analytic construction needs a table without LayerNorm")
    out = {"latent_dim": L, "use_tanh": bool(use_tanh)}
    an = np.asarray(aniso, dtype=np.float64)
    for l, (od, idim) in enumerate(shp):
        if analytic:
            if l == 0:
                u = rs.randn(od, 3)
                u /= np.linalg.norm(u, axis=1, keepdims=True)
                omega = freq_sigma * rs.randn(L, 3)
                psi = rs.uniform(0, 2 * np.pi, L)
                w = np.concatenate([latent_gain / np.sqrt(L) * np.sqrt(2.0) * np.cos(u @ omega.T + psi), u * an], axis=1)
            else:
                prev = shp[l - 1][0]                      # the part of the input that is the previous layer's output
                w = np.zeros((od, idim))
                w[:, :prev] = (4.0 if l == n - 1 else 1.0) / prev
                w += noise * rs.randn(od, idim) / idim
            b = np.zeros(od)
            if l == n - 1:
                b[0] = -r0
        else:
            w = rs.randn(od, idim) * np.sqrt(2.0 / idim)
            b = 0.1 * rs.randn(od)
            if l == n - 1:
                w *= 0.25
        w = w.astype(np.float32)
        if weight_norm and l in norm_layers:
            g = np.linalg.norm(w.astype(np.float64), axis=1, keepdims=True) * (1.0 + 0.05 * rs.randn(od, 1))
            out[f"lin{l}.weight_v"] = w
            out[f"lin{l}.weight_g"] = g.astype(np.float32)
        else:
            out[f"lin{l}.weight"] = w
        out[f"lin{l}.bias"] = b.astype(np.float32)
        if has_ln[l]:
            out[f"bn{l}.weight"] = (1.0 + 0.2 * rs.randn(od)).astype(np.float32)
            out[f"bn{l}.bias"] = (0.1 * rs.randn(od)).astype(np.float32)
    return out
