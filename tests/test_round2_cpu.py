"""CPU tests of the round-2 fixtures (generated from the imported reference by tests/golden/make_golden_r2.py):
G13 real-checkpoint ingest (host side: `module.` prefix, weight-norm fold, latent file), G14 PrecisionRecall curves /
AUC, G15 get_pose_init / T_wo initialisation / final-pose outlier rule.  No GPU, no libhortihip calls."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as GU

G13_DECODER = dict(latent_dim=32, seed=7, r0=0.04, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05, bias_sigma=0.01)
SPECS = {"NetworkArch": "deep_sdf_decoder", "CodeLength": 32,
         "NetworkSpecs": {"dims": [512] * 8, "dropout": list(range(8)), "dropout_prob": 0.2,
                          "norm_layers": list(range(8)), "latent_in": [4], "xyz_in_all": False, "use_tanh": False,
                          "latent_dropout": False, "weight_norm": True}}


def write_experiment_dir(d, params, codes, prefix="module."):
    """An experiment directory as DeepSDF's training writes it (deepsdf/deep_sdf/workspace.py:203-225, 82-114)."""
    os.makedirs(os.path.join(d, "ModelParameters"), exist_ok=True)
    os.makedirs(os.path.join(d, "LatentCodes"), exist_ok=True)
    json.dump(SPECS, open(os.path.join(d, "specs.json"), "w"))
    sd = {prefix + k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items() if k not in ("latent_dim", "hidden")}
    torch.save({"epoch": 3000, "model_state_dict": sd}, os.path.join(d, "ModelParameters", "latest.pth"))
    torch.save({"epoch": 3000, "latent_codes": {"weight": torch.from_numpy(codes)}},
               os.path.join(d, "LatentCodes", "latest.pth"))
    return sorted(sd.keys())


def test_g13_checkpoint_keys_fold_and_latents(tmp_path):
    from hortimapping_amd import synthetic as S
    from hortimapping_amd.decoder import fold_state_dict, load_latent_vectors
    g = GU.load("g13_checkpoint_ingest")
    params = S.make_synthetic_decoder(**G13_DECODER)
    keys = write_experiment_dir(str(tmp_path), params, g["codes"])
    assert keys == [str(k) for k in g["state_keys"]]                     # the DataParallel key set of the reference
    saved = torch.load(os.path.join(str(tmp_path), "ModelParameters", "latest.pth"), map_location="cpu")
    Ws, bs = fold_state_dict(saved["model_state_dict"])
    for l in range(8):
        rows = Ws[l][[0, 5, Ws[l].shape[0] - 1]]
        assert GU.relmax(rows, g[f"W{l}_rows"]) < 1e-6   # torch._weight_norm of the reference module (norm summed in another order)
        assert abs(Ws[l].astype(np.float64).sum() - float(g[f"W{l}_sum"])) < 1e-4 * max(1.0, abs(float(g[f"W{l}_sum"])))
    lat = load_latent_vectors(str(tmp_path))
    assert tuple(lat.shape) == (11, 32)
    assert np.array_equal(torch.mean(lat, dim=0).numpy(), g["init_latent"])   # test_wild_completion.py:46-47
    # the oracle on the folded weights reproduces the reference module's outputs
    from oracle import hm_oracle as O
    od = O.FoldedDecoder([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], 32)
    y = O.decoder_forward(od, torch.from_numpy(g["z"]), torch.from_numpy(g["x"])).numpy()
    assert GU.relmax(y, g["y"]) < 5e-6


def test_layernorm_checkpoints_are_refused():
    from hortimapping_amd.decoder import fold_state_dict
    sd = {"module.lin0.weight": np.zeros((512, 35), np.float32), "module.bn0.weight": np.ones(512, np.float32)}
    with pytest.raises(NotImplementedError, match="LayerNorm"):
        fold_state_dict(sd)


def test_g14_precision_recall_curves_and_auc():
    from hortimapping_amd.metrics import PrecisionRecall
    g = GU.load("g14_precision_recall")
    pr = PrecisionRecall(0.001, 0.01, 100)
    assert np.array_equal(pr.thresholds, g["thresholds"])
    import hortimapping_amd.metrics as MX
    for i in range(3):
        d_pg, d_gp = g[f"d_pg_{i}"], g[f"d_gp_{i}"]
        # feed the class the same nearest-neighbour distances the reference saw (its Open3D query is the backend)
        orig = MX._nn
        calls = iter([d_pg, d_gp])
        MX._nn = lambda a, b, backend="kdtree": next(calls)
        try:
            pr.update(np.zeros((len(d_gp), 3)), np.zeros((len(d_pg), 3)))
        finally:
            MX._nn = orig
    p_all, r_all, f_all = pr.compute_at_all_thresholds()
    assert np.allclose(p_all, g["pr_all"], rtol=0, atol=1e-12) and np.allclose(r_all, g["re_all"], rtol=0, atol=1e-12)
    assert np.allclose(f_all, g["f1_all"], rtol=0, atol=1e-12)
    assert np.allclose(np.array(pr.compute_at_threshold(0.005), dtype=np.float64), g["at5"], rtol=0, atol=1e-12)
    assert np.allclose(np.array(pr.compute_auc()), g["auc"], rtol=1e-12, atol=0)


def test_g15_pose_init_T_wo_init_and_outlier_rule():
    from hortimapping_amd import data_prep as DP
    g = GU.load("g15_pose_handling")
    for i in range(5):
        c, rot, bbx, valid = DP.get_pose_init(g[f"pts_{i}"], g[f"bg_{i}"])
        ref = g[f"pose_init_{i}"]
        assert bool(ref[5]) == valid and abs(bbx - ref[4]) < 1e-12
        if valid:
            assert np.allclose(c, ref[:3], rtol=0, atol=1e-12) and abs(rot - ref[3]) < 1e-12
    for k in range(int(g["n_init"])):
        cx, cy, cz, rot, bbx, rot_on, scale_on, r_max = g[f"init_in_{k}"]
        cfg_opt = {"pose_init": {"rot_on": bool(rot_on), "scale_on": bool(scale_on)}}
        T_wo = DP.init_T_wo(np.array([cx, cy, cz]), rot, bbx, cfg_opt, r_max)
        assert np.abs(T_wo - g[f"init_T_wo_{k}"]).max() < 2e-7            # the reference builds it in fp32
        assert np.abs(np.linalg.inv(T_wo) - g[f"init_T_ow_{k}"]).max() < 5e-6
    outl = {"scale_max": 1.25, "scale_min": 0.5, "rot_max_deg": 60}
    kept = 0
    for T_ow, ref in zip(g["outlier_T_ow"], g["outlier_res"]):
        _, s, (yaw, pitch, roll), keep = DP.final_pose_check(T_ow, outl)
        assert abs(s - ref[0]) < 1e-12 and np.allclose([yaw, pitch, roll], ref[1:4], atol=1e-9)
        assert keep == bool(ref[4])
        kept += keep
    assert 0 < kept < len(g["outlier_res"])                               # both branches exercised


def test_voxel_down_sample_matches_definition():
    from hortimapping_amd import data_prep as DP
    rs = np.random.RandomState(0)
    p = rs.rand(2000, 3) * 0.05
    q = DP.voxel_down_sample(p, 0.005)
    origin = p.min(0) - 0.0025
    idx = np.floor((p - origin) / 0.005).astype(int)
    keys = {tuple(i) for i in idx}
    assert len(q) == len(keys)
    k0 = tuple(idx[0])
    m = np.all(idx == np.array(k0), axis=1)
    assert np.abs(q - p[m].mean(0)).sum(axis=1).min() < 1e-12


def test_marching_cubes_base_case_table():
    """oracle/level_set.py's 256 -> base-case table: class sizes of the 14 classes (complement and mirror merged) and a
    few hand-checked configurations; used by tests/test_gpu_mesher.py to bound where Lewiner's MC33 adds a vertex."""
    import numpy as np
    from oracle import level_set as LS
    t = LS.mc_case_table()
    assert np.bincount(t, minlength=14).tolist() == [2, 16, 24, 24, 8, 48, 48, 16, 6, 8, 6, 24, 24, 2]
    assert t[0b00000001] == 1 and t[0b00000011] == 2 and t[0b00001001] == 3 and t[0b10000001] == 4
    assert t[0b01101001] == 13 and t[0b10010110] == 13 and t[0b00001111] == 8 and t[0b11111110] == 1
    vol = np.ones((3, 3, 3)); vol[1, 1, 1] = -1.0                           # one inside corner shared by the 8 cells
    h = LS.mc_case_histogram(vol, 0.0)
    assert h[1] == 8 and h.sum() == 8
