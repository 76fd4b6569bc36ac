"""Synthetic workloads of BASELINE.json / SURVEY.md 8d (data generators; nothing here is timed)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from . import ops, synthetic as S
from .decoder import DecoderWeights
from .optimizer import Instance

# the `opt:` block of configs/wild_pepper.yaml (values), benchmark overrides applied by c2_opt_cfg()
WILD_PEPPER_OPT = {
    "scale_on": True,
    "lm": {"lm_on": True, "lm_eye": False, "lm_lambda_0": 0.1, "s_damp": 1e-3},
    "pose_init": {"rot_on": True, "scale_on": True},
    "recon": {"n_pts": 2000, "cluster_dist_m": 0.01, "robust_th_m": 0.01},
    "render": {"n_fg_pix": 200, "n_bg_pix": 200, "n_bg_pad": 20, "n_frame": 10, "n_sample_on_ray": 30,
               "log_sdf_occ": True, "occ_cutoff_m": 0.01, "occlusion_on": True, "robust_th_m": 0.05},
    "weight": {"w_recon": 1.0, "w_depth": 5e-2, "w_mask": 5e-4, "w_codereg": 5e-4},
    "converge": {"max_iter": 50, "epsilon_g": 1e-4, "epsilon_c": 1e-2, "epsilon_t": 1e-3, "epsilon_r": 1.0,
                 "epsilon_s": 1e-3},
    "robust_iter": 5,
    "outlier": {"scale_max": 1.25, "scale_min": 0.5, "rot_max_deg": 60},
}


def c2_opt_cfg(max_iter=200, n_sample_on_ray=16, n_frame=1):
    """C2: wild_pepper weights, 200 forced iterations (all epsilon = 0), robust_iter 5, Sim(3)."""
    o = copy.deepcopy(WILD_PEPPER_OPT)
    o["converge"].update(max_iter=max_iter, epsilon_g=0.0, epsilon_c=0.0, epsilon_t=0.0, epsilon_r=0.0, epsilon_s=0.0)
    o["render"].update(n_sample_on_ray=n_sample_on_ray, n_frame=n_frame)
    return o


# The WELL-CONDITIONED full-size case: same sizes as C2 (L = 256, 8 x 512 decoder, 200 forced
# iterations, free Sim(3) pose), chosen by MEASURING the reference algorithm's own response to one-ulp input changes
# (scripts/find_wellconditioned.py; profiles/r03_wc_search*.txt) until BASELINE.json's "within 1e-4 relative" can be
# tested outright: a more elongated fruit (decoder anisotropy 0.6 : 1 : 2, so that the rotation is observable -- on the
# bench's 1 : 0.75 : 1.3 fruit the rotation error alone moves by more than 1e-4 of itself), a 4-frame render block
# (4 x 128 rays x 16 samples, 8 cm camera baseline), the render terms weighted 10 x lower (their hard sample-set switches
# are what makes the C2 iteration chaotic) and lm_lambda_0 = 1.0 (the value of lab_berry.yaml); the code regulariser keeps
# wild_pepper's 5e-4, so the latent moves as far as in C2 (a stronger one would be more stable still but pins the latent at
# zero).  All of these are user-settable YAML values of the reference's config schema.  Even so only about one candidate
# instance in six is stable to a third of the tolerance in ALL four metrics (the rotation error is the sensitive one):
# the fixture keeps those (scripts/select_wc_instances.py, tests/golden/wc_selection.json).
WC_DECODER_KW = dict(seed=2, r0=0.04, aniso=(0.6, 1.0, 2.0))
WC_INSTANCE_KW = dict(n_pts=1024, n_frames=4, n_fg=64, n_bg=64, baseline=0.08)


def wc_decoder_params(latent_dim=256):
    return S.make_synthetic_decoder(latent_dim, **WC_DECODER_KW)


def wc_opt_cfg(max_iter=200):
    o = c2_opt_cfg(max_iter=max_iter, n_sample_on_ray=16, n_frame=4)
    o["weight"].update(w_depth=5e-3, w_mask=5e-5)
    o["lm"]["lm_lambda_0"] = 1.0
    return o


def make_wc_instances(params, dec, ids, device="cuda"):
    Ws, bs = S.fold_weight_norm(params)
    fac = gpu_sdf_factory(dec, device) if (dec is not None and torch.cuda.is_available()) else None
    return [S.make_instance(Ws, bs, int(params["latent_dim"]), i, sdf_fn_factory=fac, **WC_INSTANCE_KW) for i in ids]


def gpu_sdf_factory(dec: DecoderWeights, device="cuda"):
    """sdf(x) callables backed by hm_decode_batch -- used only to *generate* synthetic observations quickly."""
    def factory(z_true):
        zt = torch.from_numpy(np.asarray(z_true, dtype=np.float32)).to(device)[None].contiguous()

        def f(p):
            p = np.asarray(p, dtype=np.float32).reshape(-1, 3)
            n = p.shape[0]
            npad = (n + 63) // 64 * 64
            pts4 = torch.zeros(1, npad, 4, device=device)
            pts4[0, :n, :3] = torch.from_numpy(p).to(device)
            y, _ = ops.decode_batch(dec, zt, pts4, torch.tensor([n], dtype=torch.int32, device=device), mode=0)
            return y[0, :n].double().cpu().numpy()
        return f
    return factory


def trained_z_true(params, i):
    """Generating latent of instance i for a TRAINED decoder (`scripts/train_synthetic_deepsdf.py`): a point a quarter
    of the way between two of its learnt codes, picked by a per-instance stream -- a plausible shape that is none of the
    training shapes and (unlike a midpoint) not close to the mean shape the optimisation starts from."""
    codes = np.asarray(params["codes"], dtype=np.float32)
    a, b = np.random.RandomState(3000 + int(i)).choice(codes.shape[0], 2, replace=False)
    return 0.75 * codes[a] + 0.25 * codes[b]


def make_c2_instances(params, dec, ids, kind="joint", device="cuda"):
    """C2-joint: 1024 surface points + 1 frame x (32 fg + 32 bg) rays; C2-sdf: 2048 surface points, no rays; "joint2048":
    the literal reading of BASELINE.json's "2048 pts/instance" for the JOINT loop -- 2048 surface points plus the same
    64-ray render block (VERDICT r04 missing #4).  A params dict that carries learnt `codes` (a trained decoder) draws
    its generating latents from them."""
    Ws, bs = S.fold_weight_norm(params)
    L = int(params["latent_dim"])
    fac = gpu_sdf_factory(dec, device) if (dec is not None and torch.cuda.is_available()) else None
    out = []
    for i in ids:
        zt = trained_z_true(params, i) if "codes" in params else None
        if kind == "joint":
            d = S.make_instance(Ws, bs, L, i, n_pts=1024, n_frames=1, n_fg=32, n_bg=32, sdf_fn_factory=fac, z_true=zt)
        elif kind == "joint2048":
            d = S.make_instance(Ws, bs, L, i, n_pts=2048, n_frames=1, n_fg=32, n_bg=32, sdf_fn_factory=fac, z_true=zt)
        else:
            d = S.make_instance(Ws, bs, L, i, n_pts=2048, n_frames=1, n_fg=4, n_bg=4, sdf_fn_factory=fac, z_true=zt)
        if zt is not None:                       # start from the mean learnt code, as `test_wild_completion.py:46-47` does
            d["latent0"] = np.asarray(params["codes"], dtype=np.float32).mean(0)
        out.append(d)
    return out


def to_instance(d, pose_known=False) -> Instance:
    rd = {k: [torch.from_numpy(a) for a in v] for k, v in d["render"].items()}
    return Instance(torch.from_numpy(d["latent0"].copy()), torch.from_numpy(d["T_ow0"].copy()),
                    torch.from_numpy(d["points_w"]), rd, float(d["cube_radius"]), pose_known)


def fixture_dicts(inp):
    """Instance dicts (the layout of `synthetic.make_instance`) from an inputs fixture under tests/golden/: arrays
    stacked over instances; render arrays carry a frame axis when the fixture has `n_frames` (multi-frame cases) and none
    otherwise (the one-frame C2 fixture)."""
    keys = ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg")
    out = []
    for k in range(inp["latent0"].shape[0]):
        if "n_frames" in getattr(inp, "files", inp):
            rd = {key: [np.asarray(inp[key][k][f]) for f in range(int(inp["n_frames"][k]))] for key in keys}
        else:
            rd = {key: [np.asarray(inp[key][k])] for key in keys}
        d = {"latent0": np.asarray(inp["latent0"][k]), "T_ow0": np.asarray(inp["T_ow0"][k]),
             "points_w": np.asarray(inp["points_w"][k]), "render": rd, "cube_radius": float(inp["cube_radius"][k])}
        for opt_key in ("z_true", "T_wo_true"):
            if opt_key in getattr(inp, "files", inp):
                d[opt_key] = np.asarray(inp[opt_key][k])
        out.append(d)
    return out
