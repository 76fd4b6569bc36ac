#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the ACTUAL reference code on CPU.

Run in the build container only (needs the read-only mount /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference has no tests and no golden vectors of its own, so these fixtures --
inputs + outputs of the reference's own functions -- are the parity pin for `oracle/hm_oracle.py`
and, through it, for the HIP path.  Decoder weights are NOT stored: they are regenerated from a seed
by `hortimapping_amd.synthetic.make_synthetic_decoder` (np.random.RandomState, frozen stream); every
other input is stored explicitly, so the fixtures are pure data.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_shim                      # noqa: E402
from oracle import hm_oracle as O                # noqa: E402  (only for default_opt_cfg)
from hortimapping_amd import synthetic as S      # noqa: E402

DEC_SPECS = {
    # name: kwargs for make_synthetic_decoder
    "pepper32": dict(latent_dim=32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05),
    "pepper256": dict(latent_dim=256, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05),
    "berry32": dict(latent_dim=32, seed=3, r0=0.02, aniso=(1.0, 1.2, 0.9), wn_perturb=0.0),
}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def mkcfg(max_iter, **kw):
    o = O.default_opt_cfg()
    o = copy.deepcopy(o)
    o["converge"]["max_iter"] = max_iter
    for k in ("epsilon_g", "epsilon_c", "epsilon_t", "epsilon_r", "epsilon_s"):
        o["converge"][k] = kw.get(k, 0.0)
    o["scale_on"] = kw.get("scale_on", True)
    o["lm"]["lm_eye"] = kw.get("lm_eye", False)
    o["lm"]["lm_on"] = kw.get("lm_on", True)
    o["lm"]["lm_lambda_0"] = kw.get("lm_lambda_0", 0.1)
    o["lm"]["s_damp"] = kw.get("s_damp", 1e-3)
    o["robust_iter"] = kw.get("robust_iter", 5)
    for k in ("n_sample_on_ray", "log_sdf_occ", "occ_cutoff_m", "occlusion_on", "n_frame"):
        if k in kw:
            o["render"][k] = kw[k]
    for k in ("w_recon", "w_depth", "w_mask", "w_codereg"):
        if k in kw:
            o["weight"][k] = kw[k]
    if "recon_robust_th_m" in kw:
        o["recon"]["robust_th_m"] = kw["recon_robust_th_m"]
    if "render_robust_th_m" in kw:
        o["render"]["robust_th_m"] = kw["render_robust_th_m"]
    return {"device": "cpu", "opt": o, "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}


def flat_cfg(cfg):
    """opt cfg -> flat dict of scalars storable in an npz."""
    o = cfg["opt"]
    out = {"scale_on": o["scale_on"], "robust_iter": o["robust_iter"]}
    for sec in ("lm", "recon", "render", "weight", "converge"):
        for k, v in o[sec].items():
            out[f"{sec}.{k}"] = v
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def inst_arrays(inst):
    d = {"points_w": inst["points_w"], "T_ow0": inst["T_ow0"], "latent0": inst["latent0"],
         "n_frames": np.int32(len(inst["render"]["T_wc"])), "cube_radius": np.float32(inst["cube_radius"])}
    for f in range(len(inst["render"]["T_wc"])):
        for k in ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg"):
            d[f"{k}_{f}"] = inst["render"][k][f]
    return d


def render_dict(inst):
    return {k: [t(a) for a in v] for k, v in inst["render"].items()}


def main():
    ns = ref_shim.import_reference()
    rs = np.random.RandomState(12345)
    decs, folded = {}, {}
    for name, kw in DEC_SPECS.items():
        p = S.make_synthetic_decoder(**kw)
        decs[name] = ref_shim.build_reference_decoder(ns, p)
        folded[name] = S.fold_weight_norm(p)

    # ---------------- G1/G2: decoder forward + per-query Jacobian ----------------
    for name in ("pepper32", "pepper256"):
        L = DEC_SPECS[name]["latent_dim"]
        z = (0.07 * rs.randn(L)).astype(np.float32)
        x = (0.05 * rs.randn(64, 3)).astype(np.float32)
        sdf = ns.utils.decode_sdf(decs[name], t(z), t(x)).numpy()
        y, g = ns.utils.get_batch_sdf_jacobian(decs[name], t(z), t(x))
        save(f"g12_decoder_{name}", decoder=name, z=z, x=x, sdf=sdf, y=y.numpy().reshape(-1),
             g=g.numpy().reshape(64, L + 3))

    # ---------------- G3: pose Jacobians ----------------
    pts = (0.05 * rs.randn(16, 3)).astype(np.float32)
    save("g3_pose_jac", pts=pts, se3=ns.utils.get_points_to_pose_jacobian_se3(t(pts)).numpy(),
         sim3=ns.utils.get_points_to_pose_jacobian_sim3(t(pts)).numpy())

    # ---------------- G4: exp maps (incl. quirk cases) ----------------
    tang = (0.1 * rs.randn(16, 7)).astype(np.float32)
    tang[0, 3:6] = 0; tang[0, 6] = 0.0            # theta = 0, s = 0
    tang[1, 3:6] = 0; tang[1, 6] = 0.05           # theta = 0, s > 0
    tang[2, 3:6] = 0; tang[2, 6] = -0.05          # theta = 0, s < 0
    tang[3, 6] = -0.05                            # theta > 0, s < 0  -> c = 0 quirk
    tang[4, 6] = 0.0                              # theta > 0, s = 0  -> c = 0 quirk
    tang[5, 3:6] = np.array([1e-9, 0, 0])         # theta ~ 1e-9
    tang[6, 3:6] = np.array([2.0, -1.5, 1.0])     # large theta
    tang[7, 6] = 1e-9                             # s below eps
    sim3 = np.stack([ns.utils.exp_sim3(t(v)).numpy() for v in tang])
    se3 = np.stack([ns.utils.exp_se3(t(v[:6])).numpy() for v in tang])
    save("g4_exp", tangents=tang, sim3=sim3, se3=se3)

    # ---------------- G5: Huber ----------------
    r = (0.02 * rs.randn(64)).astype(np.float32)
    r[:4] = 0.0
    r[4] = 0.01; r[5] = -0.01
    rr, w2 = ns.utils.get_robust_res(t(r.copy()), 0.01)
    save("g5_huber", res=r, b=np.float32(0.01), robust_res=rr.numpy().reshape(-1), w2=w2.numpy().reshape(-1))

    # ---------------- instances used below ----------------
    insts = {}
    for name, kw in (("pepper32", dict(n_pts=256, n_frames=2, n_fg=100, n_bg=100, r_max=0.08)),
                     ("pepper256", dict(n_pts=128, n_frames=1, n_fg=48, n_bg=48, r_max=0.08)),
                     ("berry32", dict(n_pts=256, n_frames=2, n_fg=100, n_bg=60, r_max=0.04))):
        Ws, bs = folded[name]
        insts[name] = S.make_instance(Ws, bs, DEC_SPECS[name]["latent_dim"], inst_id=7, **kw)

    # ---------------- G6: compute_sdf_loss ----------------
    for name in ("pepper32", "pepper256"):
        inst = insts[name]
        L = DEC_SPECS[name]["latent_dim"]
        z = (0.05 * rs.randn(L)).astype(np.float32)
        T = t(inst["T_ow0"])
        pw = t(inst["points_w"])
        pts_o = ((pw[..., None, :] * T[:3, :3]).sum(-1) + T[:3, 3])
        out = {"decoder": name, "z": z, "pts_o": pts_o.numpy()}
        for so in (True, False):
            res, jp, jc = ns.loss.compute_sdf_loss(decs[name], t(z), pts_o, so)
            sfx = "sim3" if so else "se3"
            out[f"res_{sfx}"] = res.numpy().reshape(-1)
            out[f"J_pose_{sfx}"] = jp.numpy()[:, 0]
            out[f"J_code_{sfx}"] = jc.numpy()[:, 0]
        save(f"g6_sdf_loss_{name}", **out)

    # ---------------- G7: compute_render_loss ----------------
    render_cases = [
        ("wild", "pepper32", dict(log=True, occl=True, M=30, th=0.01, scale_on=True)),
        ("lab", "pepper32", dict(log=False, occl=False, M=20, th=0.005, scale_on=False)),
        ("berry", "berry32", dict(log=True, occl=False, M=15, th=0.005, scale_on=True)),
        ("wild256", "pepper256", dict(log=True, occl=True, M=16, th=0.01, scale_on=True)),
    ]
    for cname, dname, c in render_cases:
        inst = insts[dname]
        L = DEC_SPECS[dname]["latent_dim"]
        z = (0.05 * rs.randn(L)).astype(np.float32)
        rd = render_dict(inst)
        T_ow = t(inst["T_ow0"])
        out = {"decoder": dname, "z": z, "log_occ_on": c["log"], "occlusion_on": c["occl"],
               "occupancy_th": np.float32(c["th"]), "scale_on": c["scale_on"],
               "n_frames": np.int32(len(rd["T_wc"]))}
        for f in range(len(rd["T_wc"])):
            T_oc = T_ow @ rd["T_wc"][f]
            T_co = torch.inverse(T_oc)
            rho = torch.tensor(inst["cube_radius"])
            sd = torch.linspace(T_co[2, 3] - rho, T_co[2, 3] + 0.8 * rho, c["M"])
            rays = torch.cat([rd["rays_fg"][f], rd["rays_bg"][f]], 0)
            rr = ns.loss.compute_render_loss(decs[dname], t(z), rays, rd["depth_fg"][f], rd["depth_bg"][f],
                                             T_oc, sd, c["scale_on"], c["log"], c["th"], float(rho), c["occl"])
            out.update({f"rays_{f}": rays.numpy(), f"depth_fg_{f}": rd["depth_fg"][f].numpy(),
                        f"depth_bg_{f}": rd["depth_bg"][f].numpy(), f"T_oc_{f}": T_oc.numpy(),
                        f"sampled_depth_{f}": sd.numpy(), f"bbx_radius_{f}": np.float32(rho)})
            out.update({f"res_d_{f}": rr[0].numpy().reshape(-1), f"J_d_pose_{f}": rr[1].numpy()[:, 0],
                        f"J_d_code_{f}": rr[2].numpy()[:, 0], f"res_m_{f}": rr[3].numpy().reshape(-1),
                        f"J_m_pose_{f}": rr[4].numpy()[:, 0], f"J_m_code_{f}": rr[5].numpy()[:, 0]})
        save(f"g7_render_{cname}", **out)
    # None case: ball radius so small that < 100 samples are valid (loss.py:43-45)
    inst = insts["pepper32"]
    rd = render_dict(inst)
    T_oc = t(inst["T_ow0"]) @ rd["T_wc"][0]
    T_co = torch.inverse(T_oc)
    sd = torch.linspace(T_co[2, 3] - 0.08, T_co[2, 3] + 0.064, 30)
    rays = torch.cat([rd["rays_fg"][0], rd["rays_bg"][0]], 0)
    z = (0.05 * rs.randn(32)).astype(np.float32)
    rr = ns.loss.compute_render_loss(decs["pepper32"], t(z), rays, rd["depth_fg"][0], rd["depth_bg"][0], T_oc, sd,
                                     True, True, 0.01, 0.004, True)
    assert rr is None
    save("g7_render_none", decoder="pepper32", z=z, rays_0=rays.numpy(), depth_fg_0=rd["depth_fg"][0].numpy(),
         depth_bg_0=rd["depth_bg"][0].numpy(), T_oc_0=T_oc.numpy(), sampled_depth_0=sd.numpy(),
         bbx_radius_0=np.float32(0.004), is_none=True)

    # ---------------- G8: one LM iteration H, b, delta (captured inside the reference loop) ----------------
    captured = {}
    real_inverse, real_mv = torch.inverse, torch.mv

    def cap_inverse(A):
        if A.shape[0] > 4:
            captured["H"] = A.clone()
        return real_inverse(A)

    def cap_mv(A, v):
        out = real_mv(A, v)
        if A.shape[0] > 4:          # exp_se3/exp_sim3 also call torch.mv on 3-vectors (utils.py:252,322)
            captured["b"] = v.clone()
            captured["delta"] = out.clone()
        return out

    for name in ("pepper32", "pepper256"):
        inst = insts[name]
        L = DEC_SPECS[name]["latent_dim"]
        cfg = mkcfg(1, n_sample_on_ray=(30 if L == 32 else 16))
        opt = ns.optimizer.Optimizer(cfg, decs[name], None, None)
        z0 = (0.03 * rs.randn(L)).astype(np.float32)
        out = {"decoder": name, "z0": z0, **inst_arrays(inst), **{"cfg." + k: v for k, v in flat_cfg(cfg).items()}}
        for tag, pk in (("free", False),):
            torch.inverse, torch.mv = cap_inverse, cap_mv
            try:
                zr, Tr, nr = opt.shape_pose_joint_opt(t(z0.copy()), t(inst["T_ow0"]), render_dict(inst),
                                                      t(inst["points_w"]), inst["cube_radius"], None, pose_known=pk)
            finally:
                torch.inverse, torch.mv = real_inverse, real_mv
            out.update({f"H_{tag}": captured["H"].numpy(), f"b_{tag}": captured["b"].numpy(),
                        f"delta_{tag}": captured["delta"].numpy(), f"z_{tag}": zr.numpy(), f"T_{tag}": Tr.numpy()})
        # shape-only loop
        torch.inverse, torch.mv = cap_inverse, cap_mv
        try:
            zr, Tr, nr = opt.shape_opt_deepsdf(t(z0.copy()), t(inst["T_ow0"]), t(inst["points_w"]), None)
        finally:
            torch.inverse, torch.mv = real_inverse, real_mv
        out.update({"H_sdf": captured["H"].numpy(), "b_sdf": captured["b"].numpy(),
                    "delta_sdf": captured["delta"].numpy(), "z_sdf": zr.numpy()})
        save(f"g8_one_iter_{name}", **out)

    # ---------------- G9: trajectories ----------------
    traj_cases = []
    for it in (1, 2, 5, 20):
        traj_cases.append((f"known_sim3_it{it}", "pepper32", dict(max_iter=it), True, "joint"))
        traj_cases.append((f"sdf_it{it}", "pepper32", dict(max_iter=it), False, "sdf"))
    for it in (1, 2, 5):
        traj_cases.append((f"free_sim3_it{it}", "pepper32", dict(max_iter=it), False, "joint"))
    traj_cases += [
        ("known_se3_it5", "pepper32", dict(max_iter=5, scale_on=False), True, "joint"),
        ("free_se3_it2", "pepper32", dict(max_iter=2, scale_on=False), False, "joint"),
        ("known_lmeye_it5", "pepper32", dict(max_iter=5, lm_eye=True), True, "joint"),
        ("known_gn_it3", "pepper32", dict(max_iter=3, lm_on=False), True, "joint"),
        ("known_lab_it5", "pepper32", dict(max_iter=5, scale_on=False, log_sdf_occ=False, occlusion_on=False,
                                            n_sample_on_ray=20, occ_cutoff_m=0.005, robust_iter=1,
                                            w_recon=0.01, w_mask=1e-3, w_codereg=1e-3,
                                            recon_robust_th_m=0.005, render_robust_th_m=0.02), True, "joint"),
        ("known_berry_it5", "berry32", dict(max_iter=5, n_sample_on_ray=15, occ_cutoff_m=0.005, occlusion_on=False,
                                             lm_lambda_0=1.0, s_damp=0.0, recon_robust_th_m=0.003,
                                             robust_iter=100), True, "joint"),
        ("known_256_it3", "pepper256", dict(max_iter=3, n_sample_on_ray=16), True, "joint"),
        ("sdf_256_it5", "pepper256", dict(max_iter=5), False, "sdf"),
        # early exits
        ("exit_grad", "pepper32", dict(max_iter=30, epsilon_g=2e-2), True, "joint"),
        ("exit_grad_free", "pepper32", dict(max_iter=30, epsilon_g=1e-4), False, "joint"),
        ("sdf_exit_grad", "pepper32", dict(max_iter=30, epsilon_g=1e-6), False, "sdf"),
        ("exit_code", "pepper32", dict(max_iter=30, epsilon_c=2.0), True, "joint"),
        ("sdf_exit_code", "pepper32", dict(max_iter=30, epsilon_c=5e-2), False, "sdf"),
    ]
    for tag, dname, kw, pk, kind in traj_cases:
        inst = insts[dname]
        cfg = mkcfg(**kw)
        opt = ns.optimizer.Optimizer(cfg, decs[dname], None, None)
        z0 = t(inst["latent0"].copy())
        if kind == "joint":
            zr, Tr, nr = opt.shape_pose_joint_opt(z0, t(inst["T_ow0"]), render_dict(inst), t(inst["points_w"]),
                                                  inst["cube_radius"], None, pose_known=pk)
        else:
            zr, Tr, nr = opt.shape_opt_deepsdf(z0, t(inst["T_ow0"]), t(inst["points_w"]), None)
        print(tag, "iter_count", nr)
        save(f"g9_traj_{tag}", decoder=dname, kind=kind, pose_known=pk, **inst_arrays(inst),
             **{"cfg." + k: v for k, v in flat_cfg(cfg).items()},
             z_out=zr.numpy(), T_out=Tr.numpy(), iter_count=np.int32(nr))
    # invalid submap (optimizer.py:139-141): (a) object shifted sideways so that no ray sample falls in the ball ->
    # every frame returns None -> 0 depth observations -> break at i=0; (b) shifted along the optical axis -> the
    # loop runs a few iterations and then loses all depth observations -> break with iter_count = i.
    for tag, shift in (("invalid_at0", [0.35, 0.0, 0.0]), ("invalid_later", [0.0, 0.0, 0.35])):
        inst = copy.deepcopy(insts["pepper32"])
        T_bad = inst["T_ow0"].copy(); T_bad[:3, 3] += np.array(shift, dtype=np.float32)
        inst["T_ow0"] = T_bad
        cfg = mkcfg(8)
        opt = ns.optimizer.Optimizer(cfg, decs["pepper32"], None, None)
        zr, Tr, nr = opt.shape_pose_joint_opt(t(inst["latent0"].copy()), t(T_bad), render_dict(inst),
                                              t(inst["points_w"]), inst["cube_radius"], None, pose_known=False)
        print(tag, "iter_count", nr)
        save(f"g9_traj_{tag}", decoder="pepper32", kind="joint", pose_known=False, **inst_arrays(inst),
             **{"cfg." + k: v for k, v in flat_cfg(cfg).items()}, z_out=zr.numpy(), T_out=Tr.numpy(),
             iter_count=np.int32(nr))

    # ---------------- G10: caller-side data prep pins (numpy-only reference functions) ----------------
    K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1.0]])
    pix = np.array([[0, 0], [320, 240], [100.5, 33.25], [639, 479], [17, 400]], dtype=np.float64)
    rays = ns.utils.get_rays(pix, np.linalg.inv(K))
    # get_render_data on a synthetic 64x64 id/depth pair, numpy RNG fixed (utils.py:39-109)
    Hh, Ww = 64, 64
    idimg = np.zeros((Hh, Ww), dtype=np.int32)
    yy, xx = np.mgrid[0:Hh, 0:Ww]
    idimg[(yy - 30) ** 2 + (xx - 34) ** 2 < 15 ** 2] = 5
    depth = np.full((Hh, Ww), 0.9, dtype=np.float32)
    depth[idimg == 5] = 0.5 + 0.001 * (xx[idimg == 5] - 34)
    depth[10:14, 10:14] = 0.0
    depth[28:31, 30:33] = 0.0          # invalid depth inside the mask
    cfg = mkcfg(1)
    cfg["opt"]["render"]["n_fg_pix"] = 60
    cfg["opt"]["render"]["n_bg_pix"] = 50
    cfg["opt"]["render"]["n_bg_pad"] = 6
    K64 = np.array([[80.0, 0, 32.0], [0, 80.0, 32.0], [0, 0, 1.0]])
    np.random.seed(42)
    rdat = ns.utils.get_render_data(5, {0: idimg, 3: idimg.T.copy()}, {0: depth, 3: depth.T.copy()},
                                    {0: np.eye(4), 3: np.eye(4) * 1.0}, (Hh, Ww), np.linalg.inv(K64), cfg,
                                    min_pix_count_match=100, max_bbx_size=300)
    out = {"pix": pix, "K": K, "rays": rays, "id_img": idimg, "depth_img": depth, "K64": K64,
           "n_fg_pix": 60, "n_bg_pix": 50, "n_bg_pad": 6, "count": np.int32(rdat["count"])}
    for f in range(rdat["count"]):
        for k in ("rays_fg", "rays_bg", "depth_fg", "depth_bg", "T_wc"):
            out[f"{k}_{f}"] = rdat[k][f].numpy()
        out[f"pix_fg_{f}"] = rdat["pix_fg"][f]
        out[f"pix_bg_{f}"] = rdat["pix_bg"][f]
    save("g10_data_prep", **out)

    # ---------------- G11: create_voxel_grid (utils.py:542-562; pure torch) ----------------
    save("g11_voxel_grid", n5=ns.utils.create_voxel_grid(5).numpy(), n8=ns.utils.create_voxel_grid(8).numpy())


if __name__ == "__main__":
    main()
