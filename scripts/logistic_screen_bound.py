#!/usr/bin/env python3
"""Upper bound of what a screening pass could skip under LOGISTIC occupancy (GPU box):

    python scripts/logistic_screen_bound.py > profiles/r06_logistic_screen_bound.txt

Under the logistic map o = sigmoid(-s / sigma), sigma = occ_cutoff / 3 * 0.55 (reference: wild_completion/loss.py:57-61,
utils.py:136-142) no sample saturates exactly, so the linear-occupancy screening of round 5 (bit-identical) has nothing
to skip.  A tolerance-level screen could skip a ball-valid ray sample whose exact sdf value cannot move any fp32 sum of
its ray:
  class A  far outside:  s >  16.6 sigma + eps  =>  o < 2^-24 (1 - o rounds to 1, o * T adds < 2^-24 per sample),
  class C  far inside:   s < -16.6 sigma - eps  =>  1 - o < 2^-24 (the sample ends the ray whatever its exact value),
  class B  behind an accumulated transmittance T_prev < 2^-40 (nothing behind it changes an fp32 sum),
with eps = 1e-3 m, the measured bound on |sdf_fp16 - sdf_f16x3| (profiles/r05_screen_eps.txt).  None of them can be a
with-grad sample (|s| < occ_cutoff = 5.45 sigma) -- class B samples can, but their de_do <= T_prev * M * delta_d / (1 - o)
fails the reference's min_grad_thre = 1e-6 test only if T_prev is that small, which is why B uses 2^-40.
The script reports, per logistic configuration, the fraction of ball-valid samples in A / B / C at the INITIAL state and
at the CONVERGED state of the HIP path (the screen's promoted fraction p = 1 - skippable), and the break-even:
a one-pass fp16 screen costs ~1/3 of the f16x3 forward it replaces, so it pays when p < ~0.67 and reaches the
>= 150 fruits/s asked for wild_pepper only when p <~ 0.35."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yaml

from hortimapping_amd import ops, optimizer as HO, synthetic as S, workloads as W
from hortimapping_amd.decoder import DecoderWeights

EPS = 1e-3
CASES = [  # name, yaml (None = the C2 block), L, pose_known, r0, instance shape
    ("configs0 wild_pepper.yaml (10 x 400 x 30)", "wild_pepper.yaml", 32, False, 0.04, dict(n_pts=2000, n_frames=10, n_fg=200, n_bg=200)),
    ("configs4 lab_berry.yaml (8 x 600 x 15)", "lab_berry.yaml", 32, False, 0.02, dict(n_pts=2000, n_frames=8, n_fg=400, n_bg=200, r_max=0.04)),
    ("cka_pepper.yaml (as wild_pepper's block)", "cka_pepper.yaml", 32, False, 0.04, dict(n_pts=2000, n_frames=10, n_fg=200, n_bg=200)),
    ("configs1 C2-joint (1 x 64 x 16, L = 256)", None, 256, False, 0.04, dict(n_pts=1024, n_frames=1, n_fg=32, n_bg=32)),
]
N_FRUIT = 16


def classify(dec, opt, insts, lat, T_ow):
    """Counts over all fruits / frames / rays at the given state: ball-valid, A, C, B (beyond A and C), with-grad."""
    M = int(opt["render"]["n_sample_on_ray"])
    th = float(opt["render"]["occ_cutoff_m"])
    sigma = th / 3 * 0.55
    cut = 16.6 * sigma + EPS
    n_frame = int(opt["render"]["n_frame"])
    tot = np.zeros(6, dtype=np.int64)       # valid, A, C, B, with-grad, with-grad inside a skipped class (must be 0)
    dev = "cuda"
    for b, inst in enumerate(insts):
        Tow = T_ow[b].double()
        cur_scale = torch.det(Tow[:3, :3]) ** (-1 / 3)
        rd = inst.render_data
        sel = HO._select_frames(len(rd["T_wc"]), n_frame)
        for f in sel:
            T_oc = Tow @ rd["T_wc"][f].double()
            T_co = torch.inverse(T_oc)
            dr = inst.cube_radius * cur_scale
            d = torch.linspace(float(T_co[2, 3] - dr), float(T_co[2, 3] + 0.8 * dr), M, dtype=torch.float64)
            rays = torch.cat([rd["rays_fg"][f], rd["rays_bg"][f]], 0).double()
            p_c = rays[:, None, :] * d[:, None]
            p_o = (p_c[..., None, :] * T_oc[:3, :3]).sum(-1) + T_oc[:3, 3]
            valid = torch.linalg.vector_norm(p_o, dim=-1) < dr
            R = rays.shape[0]
            n = R * M
            ns = (n + 63) // 64 * 64
            pts4 = torch.zeros(1, ns, 4, device=dev)
            pts4[0, :n, :3] = p_o.reshape(n, 3).float().to(dev)
            y, _ = ops.decode_batch(dec, lat[b:b + 1].contiguous().to(dev), pts4,
                                    torch.tensor([n], dtype=torch.int32, device=dev), mode=0)
            s = y[0, :n].reshape(R, M).double().cpu()
            occ = torch.where(valid, torch.sigmoid(-s / sigma), torch.zeros_like(s))
            Tacc = torch.cumprod(1 - occ, dim=-1)
            T_prev = torch.cat([torch.ones(R, 1, dtype=torch.float64), Tacc[:, :-1]], dim=1)
            A = valid & (s > cut)
            C = valid & (s < -cut)
            Bc = valid & ~A & ~C & (T_prev < 2.0 ** -40)
            wg = valid & (s.abs() < th)
            tot += np.array([int(valid.sum()), int(A.sum()), int(C.sum()), int(Bc.sum()), int(wg.sum()),
                             int((wg & (A | C)).sum())])
    return tot


def main():
    print(__doc__.split("\n\n")[0].split("\n")[0])
    print(f"eps {EPS} m; {N_FRUIT} synthetic fruits per configuration (the bench's generator), f16x3 decoder; "
          "counts over all selected frames x rays x samples\n")
    for name, y, L, known, r0, shape in CASES:
        if y is None:
            opt = W.c2_opt_cfg(max_iter=200)
            opt["converge"].update(epsilon_g=1e-4, epsilon_c=1e-2, epsilon_t=1e-3, epsilon_r=1.0, epsilon_s=1e-3)
        else:
            opt = yaml.safe_load(open(os.path.join(ROOT, "configs", y)))["opt"]
        if not opt["render"]["log_sdf_occ"]:
            print(f"{name}: linear occupancy, not a case of this table\n")
            continue
        p = S.make_synthetic_decoder(L, seed=1, r0=r0, aniso=(1.0, 0.75, 1.3))
        dec = DecoderWeights.from_params(p).set_precision("f16x3")
        Ws, bs = S.fold_weight_norm(p)
        fac = W.gpu_sdf_factory(dec)
        protos = [S.make_instance(Ws, bs, L, i, sdf_fn_factory=fac, **shape) for i in range(N_FRUIT)]
        insts = [W.to_instance(d, pose_known=known) for d in protos]
        lat0 = torch.stack([i.latent for i in insts]).float()
        T0 = torch.stack([i.T_ow for i in insts]).float()
        res = HO.optimize_batch(dec, opt, insts)
        lat1 = torch.stack([r.latent for r in res]).float()
        T1 = torch.stack([r.T_ow for r in res]).float()
        its = np.array([r.iter_count for r in res])
        th = float(opt["render"]["occ_cutoff_m"])
        print(f"{name}: occ_cutoff {th} m, sigma {th / 3 * 0.55:.5f} m, 16.6 sigma = {16.6 * th / 3 * 0.55:.4f} m, "
              f"M = {opt['render']['n_sample_on_ray']}, iterations mean {its.mean():.1f}")
        for tag, lat, T in (("initial state", lat0, T0), ("converged state", lat1, T1)):
            v, a, c, bb, wg, bad = classify(dec, opt, insts, lat, T)
            skip = a + c + bb
            print(f"  {tag:16s} ball-valid {v:9d} | A far outside {a / v:6.1%} | C far inside {c / v:6.1%} | "
                  f"B behind T < 2^-40 {bb / v:6.1%} | skippable {skip / v:6.1%} -> promoted p = {1 - skip / v:5.1%} | "
                  f"with-grad {wg / v:5.1%} (in a skipped class: {bad})")
        print()
    print("reading: the screen replaces (3 passes) by (1 pass + 3 p passes): it pays when p < 0.67, and the forward-only "
          "samples are ~85 % of wild_pepper's flop / 25 % of the C2 main launch")


if __name__ == "__main__":
    main()
