"""GPU vs oracle, iteration by iteration, on the two-frame L=256 instances whose small frame changes validity."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights
from oracle import hm_oracle as O
L = 256
p = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
od = O.fold_decoder(p)
dec = DecoderWeights.from_params(p); dec.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")
Ws, bs = S.fold_weight_norm(p)
t = torch.from_numpy
for iid in (4, 7):
    d = S.make_instance(Ws, bs, L, iid, n_pts=128, n_frames=2, n_fg=48, n_bg=48)
    for key in ("rays_fg", "rays_bg", "depth_fg", "depth_bg"):
        d["render"][key][1] = d["render"][key][1][:4]
    rd = {kk: [t(a) for a in v] for kk, v in d["render"].items()}
    for k in range(1, 9):
        opt = W.c2_opt_cfg(max_iter=k, n_sample_on_ray=16, n_frame=2)
        tr = []
        z, T, n = O.shape_pose_joint_opt(od, opt, t(d["latent0"]), t(d["T_ow0"]), rd, t(d["points_w"]), d["cube_radius"], pose_known=False, trace=tr)
        dbg = {}
        r = HO.optimize_batch(dec, opt, [W.to_instance(d, pose_known=False)], debug=dbg)[0]
        c = dbg["counts"][0].cpu().numpy()
        print(f"inst {iid} it {k}: oracle (n_valid, n_keep, V) = ({tr[-1].n_valid}, {tr[-1].n_keep}, {tr[-1].n_rays})  gpu = ({c[0]}, {c[1]}, {c[2]})  "
              f"rel latent {float((r.latent - z).abs().max() / z.abs().max()):.2e}  T {float((r.T_ow - T).abs().max()):.2e} status {r.status}", flush=True)
