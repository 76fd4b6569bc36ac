#!/usr/bin/env python3
"""Round-2 golden vectors (same rules as make_golden.py: run in the build container only, outputs are pure data).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

G13  real-checkpoint ingest: a weight-normed reference `Decoder` wrapped in DataParallel is saved the
     way the training code does ({"epoch", "model_state_dict"} with `module.` keys, specs.json, LatentCodes/latest.pth),
     re-loaded by the REFERENCE's own `config_decoder` / `load_latent_vectors` (deepsdf/deep_sdf/workspace.py:203-225,
     82-114) and evaluated; the fixture keeps the state-dict key list, probes of the folded weights and the outputs.
G14  PrecisionRecall bookkeeping (metrics_3d/precision_recall.py:58-98) on given per-update distance arrays:
     compute_at_threshold / compute_at_all_thresholds / compute_auc of the reference class itself.
G15  caller-side pose handling: `get_pose_init` (wild_completion/utils.py:420-459) on stand-in point-cloud objects, the
     T_wo initialisation (test_wild_completion.py:196-209, with the reference's axis_angle_to_rotation_matrix) and the
     final-pose outlier rule (test_wild_completion.py:228-246) evaluated on fixed inputs.
G16  perturbation noise of the reference loop itself for the short parity trajectories (tests/test_gpu_parity.py,
     tests/test_gpu_configs.py): the reference `Optimizer` re-run with the surface points scaled by (1 +- 1e-7).
"""
import copy
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_shim                      # noqa: E402
from hortimapping_amd import synthetic as S      # noqa: E402

SPECS = {"NetworkArch": "deep_sdf_decoder", "CodeLength": 32,
         "NetworkSpecs": {"dims": [512] * 8, "dropout": list(range(8)), "dropout_prob": 0.2,
                          "norm_layers": list(range(8)), "latent_in": [4], "xyz_in_all": False, "use_tanh": False,
                          "latent_dropout": False, "weight_norm": True}}
G13_DECODER = dict(latent_dim=32, seed=7, r0=0.04, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05, bias_sigma=0.01)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def g13(ns):
    import deepsdf.deep_sdf.workspace as ws
    params = S.make_synthetic_decoder(**G13_DECODER)
    dec = ref_shim.build_reference_decoder(ns, params)
    dp = torch.nn.DataParallel(dec)
    rs = np.random.RandomState(5)
    codes = (0.07 * rs.randn(11, 32)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "ModelParameters")); os.makedirs(os.path.join(d, "LatentCodes"))
        json.dump(SPECS, open(os.path.join(d, "specs.json"), "w"))
        torch.save({"epoch": 3000, "model_state_dict": dp.state_dict()}, os.path.join(d, "ModelParameters", "latest.pth"))
        torch.save({"epoch": 3000, "latent_codes": {"weight": torch.from_numpy(codes)}},
                   os.path.join(d, "LatentCodes", "latest.pth"))
        keys = sorted(torch.load(os.path.join(d, "ModelParameters", "latest.pth"))["model_state_dict"].keys())
        loaded = ws.config_decoder(d)                               # the reference's loader
        lat = ws.load_latent_vectors(d, "latest")
    init_latent = torch.mean(lat, dim=0)                            # test_wild_completion.py:46-47
    x = rs.uniform(-0.06, 0.06, (64, 3)).astype(np.float32)
    z = codes[3]
    inp = torch.cat([torch.from_numpy(z).expand(64, -1), torch.from_numpy(x)], 1)
    with torch.no_grad():
        y = loaded(inp)[:, 0].numpy()
        y_mean = loaded(torch.cat([init_latent.expand(64, -1), torch.from_numpy(x)], 1))[:, 0].numpy()
    probes = {}
    for l in range(8):
        lin = getattr(loaded, f"lin{l}")
        with torch.no_grad():
            w = torch._weight_norm(lin.weight_v, lin.weight_g, 0).numpy()
        probes[f"W{l}_rows"] = w[[0, 5, w.shape[0] - 1]]
        probes[f"W{l}_sum"] = np.float64(w.astype(np.float64).sum())
    save("g13_checkpoint_ingest", state_keys=np.array(keys), codes=codes, init_latent=init_latent.numpy(), x=x, z=z,
         y=y, y_mean=y_mean, **probes)


def g14():
    sys.modules["open3d"].geometry = types.SimpleNamespace(Geometry=type("Geometry", (), {}))
    from metrics_3d.precision_recall import PrecisionRecall
    pr = PrecisionRecall(0.001, 0.01, 100)
    rs = np.random.RandomState(3)
    d_pg = [np.abs(rs.randn(500)) * 0.004, np.abs(rs.randn(300)) * 0.002, np.abs(rs.randn(50)) * 0.02]
    d_gp = [np.abs(rs.randn(400)) * 0.003, np.abs(rs.randn(300)) * 0.006, np.abs(rs.randn(70)) * 0.0005]

    class FakePcd:                                  # stands in for the Open3D point cloud: distances are GIVEN
        def __init__(self, d): self.d = d
        def compute_point_cloud_distance(self, other): return self.d
    for a, b in zip(d_pg, d_gp):
        pr.prediction_is_empty = lambda pt: False
        pr.convert_to_pcd = staticmethod(lambda g: g)
        pr.update(FakePcd(b), FakePcd(a))           # update(gt, pt): pt.dist(gt) = precision side, gt.dist(pt) = recall
    p_all, r_all, f_all = pr.compute_at_all_thresholds()
    at5 = pr.compute_at_threshold(0.005)
    auc = pr.compute_auc()
    out = {"thresholds": pr.thresholds, "pr_all": np.array(p_all), "re_all": np.array(r_all), "f1_all": np.array(f_all),
           "at5": np.array(at5, dtype=np.float64), "auc": np.array(auc, dtype=np.float64)}
    for i, (a, b) in enumerate(zip(d_pg, d_gp)):
        out[f"d_pg_{i}"], out[f"d_gp_{i}"] = a, b
    save("g14_precision_recall", **out)


def g15(ns):
    from numpy.linalg import inv, det
    from scipy.spatial.transform import Rotation

    class Box:
        def __init__(self, lo, hi): self.lo, self.hi = np.asarray(lo, float), np.asarray(hi, float)
        def get_center(self): return 0.5 * (self.lo + self.hi)
        def get_extent(self): return self.hi - self.lo

    class Pcd:                                      # the four Open3D calls get_pose_init makes (utils.py:424,447-450)
        def __init__(self, p): self.points = np.asarray(p, float)
        def get_axis_aligned_bounding_box(self): return Box(self.points.min(0), self.points.max(0))
        def crop(self, box):
            m = np.all((self.points >= box.lo) & (self.points <= box.hi), axis=1)
            return Pcd(self.points[m])
        def __deepcopy__(self, memo): return Pcd(self.points.copy())
    sys.modules["open3d"].geometry = types.SimpleNamespace(AxisAlignedBoundingBox=Box, Geometry=type("Geometry", (), {}))
    ns.utils.o3d = sys.modules["open3d"]
    rs = np.random.RandomState(11)
    out = {}
    cases = []
    for i, (ext, bg_shift) in enumerate([((0.07, 0.09, 0.06), (0.03, 0.05, 0.09)), ((0.08, 0.05, 0.06), (-0.05, 0.02, 0.06)),
                                         ((0.2, 0.1, 0.1), (0, 0, 0.1)), ((0.01, 0.015, 0.01), (0, 0, 0.1)),
                                         ((0.06, 0.06, 0.05), None)]):
        c0 = np.array([0.1, -0.2, 0.6])
        pts = c0 + (rs.rand(400, 3) - 0.5) * np.array(ext)
        bg = (c0 + np.array(bg_shift) + 0.03 * rs.randn(300, 3)) if bg_shift is not None else c0 + 5.0 + rs.randn(20, 3)
        center, rot, bbx, valid = ns.utils.get_pose_init(Pcd(pts), Pcd(bg))
        out[f"pts_{i}"], out[f"bg_{i}"] = pts, bg
        out[f"pose_init_{i}"] = np.array([center[0], center[1], center[2], rot, bbx, float(valid)])
        cases.append((np.asarray(center), float(rot), float(bbx), bool(valid)))
    # T_wo initialisation (test_wild_completion.py:196-209) for the valid cases x the four pose_init flag settings
    r_max = 0.08
    k = 0
    for center, rot, bbx, valid in cases:
        if not valid or rot == 0.0:      # rot == 0 makes the reference divide 0/0 (utils.py:373): not a usable pin
            continue
        for rot_on, scale_on in ((True, True), (True, False)):
            T_wo = torch.eye(4, dtype=torch.float32)
            T_wo[:3, 3] = torch.tensor(center, dtype=torch.float32)
            aa = torch.tensor([0, rot if rot_on else 0.0, 0], dtype=torch.float32)
            object_radius_m = r_max * 0.8
            scale_init = max(bbx / (2 * object_radius_m), 0.5) if scale_on else 1.0
            T_wo[:3, :3] = ns.utils.axis_angle_to_rotation_matrix(aa) * scale_init
            out[f"init_in_{k}"] = np.array([center[0], center[1], center[2], rot, bbx, float(rot_on), float(scale_on), r_max])
            out[f"init_T_wo_{k}"] = T_wo.numpy()
            out[f"init_T_ow_{k}"] = torch.inverse(T_wo).numpy()
            k += 1
    out["n_init"] = np.int32(k)
    # final-pose outlier rule (test_wild_completion.py:228-246)
    outl = {"scale_max": 1.25, "scale_min": 0.5, "rot_max_deg": 60}
    T_list, res = [], []
    for i in range(12):
        ang = rs.uniform(-1.4, 1.4, 3) * (0.3 if i < 4 else 1.0)
        s = float(np.exp(rs.uniform(-0.9, 0.4)))
        T_wo = np.eye(4); T_wo[:3, :3] = s * Rotation.from_euler("zyx", ang).as_matrix(); T_wo[:3, 3] = rs.randn(3)
        T_ow_cur = inv(T_wo).astype(np.float32)
        T = inv(T_ow_cur)
        final_scale = det(T[:3, :3]) ** (1 / 3)
        yaw, pitch, roll = Rotation.from_matrix(T[:3, :3] / final_scale).as_euler("zyx", degrees=True)
        keep = not (final_scale < outl["scale_min"] or final_scale > outl["scale_max"] or abs(pitch) > outl["rot_max_deg"]
                    or abs(roll) > outl["rot_max_deg"])
        T_list.append(T_ow_cur); res.append([final_scale, yaw, pitch, roll, float(keep)])
    out["outlier_T_ow"], out["outlier_res"] = np.stack(T_list), np.array(res)
    save("g15_pose_handling", **out)


def g16(ns):
    """Noise of the reference loop for the short-trajectory parity gates: re-run every G9 case with the surface points
    scaled by (1 + 1e-7) and (1 - 1e-7) through the reference `Optimizer`; rows = the two perturbations, columns =
    (max |dz| / max |z|, max |dT| / max |T|, iter_count difference) against the unperturbed golden outputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as GU
    out, decs = {}, {}
    for name in GU.list_golden("g9_traj_"):
        g = GU.load(name)
        tag = name[len("g9_traj_"):]
        if "lin8_bias_shift" in g.files:
            continue                                   # round-5 cases: their noise rows are in g16_traj_noise_r5 (make_golden_r5.py)
        dname = str(g["decoder"])
        if dname not in decs:
            decs[dname] = ref_shim.build_reference_decoder(ns, GU.decoder_params(dname))
        cfg = {"device": "cpu", "opt": GU.cfg_from_golden(g), "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}
        devs = []
        for eps in (1e-7, -1e-7):
            opt = ns.optimizer.Optimizer(copy.deepcopy(cfg), decs[dname], None, None)
            pw = torch.from_numpy((g["points_w"] * np.float32(1 + eps)).astype(np.float32))
            z0, T0 = torch.from_numpy(g["latent0"].copy()), torch.from_numpy(g["T_ow0"].copy())
            if str(g["kind"]) == "joint":
                z, T, n = opt.shape_pose_joint_opt(z0, T0, GU.render_data_from_golden(g), pw, float(g["cube_radius"]),
                                                   None, pose_known=bool(g["pose_known"]))
            else:
                z, T, n = opt.shape_opt_deepsdf(z0, T0, pw, None)
            zr, Tr = g["z_out"], g["T_out"]
            devs.append([float(np.abs(z.numpy() - zr).max() / max(np.abs(zr).max(), 1e-12)),
                         float(np.abs(T.numpy() - Tr).max() / np.abs(Tr).max()), float(n - int(g["iter_count"]))])
        out[tag] = np.array(devs)
        print(tag, out[tag].tolist(), flush=True)
    save("g16_traj_noise", **out)


def main():
    ns = ref_shim.import_reference()
    which = sys.argv[1:] or ["g13", "g14", "g15", "g16"]
    if "g13" in which: g13(ns)
    if "g14" in which: g14()
    if "g15" in which: g15(ns)
    if "g16" in which: g16(ns)


if __name__ == "__main__":
    main()
