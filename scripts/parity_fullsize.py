"""Metric-level parity at BASELINE.json's full size (64 peppers, L=256, 200 LM iterations, free pose):
Chamfer-to-ground-truth and pose error of  (a) GPU exact-fp32  vs  (b) GPU f16x3  for all 64 instances, and of both
vs (c) the CPU oracle (fp32) and (d) the CPU oracle with inputs perturbed by 1e-7 relative (the reference's own noise
floor) on a few instances.  Prints a table; results are quoted in DESIGN.md."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO, metrics as MX, utils as U
from hortimapping_amd.decoder import DecoderWeights
from oracle import hm_oracle as O

L, B = 256, 64
n_oracle = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pose_known = len(sys.argv) > 3 and sys.argv[3] == 'known'
params = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(params)
dicts = W.make_c2_instances(params, dec, list(range(B)), kind="joint")
cfg = W.c2_opt_cfg(max_iter=iters)
insts = [W.to_instance(d, pose_known=pose_known) for d in dicts]
res = {}
for prec in ("f32", "f16x3"):
    dec.set_precision(prec)
    res[prec] = HO.optimize_batch(dec, cfg, insts)
dec.set_precision("f32")

def pts(latent, T_ow):
    return MX.completed_points_world(lambda p: U.decode_sdf(dec, latent, torch.from_numpy(p)).cpu().numpy(), np.asarray(T_ow))

def metrics(latent, T_ow, d, gt):
    cd = MX.chamfer_distance(pts(latent, T_ow), gt)
    te, re, sr = MX.pose_error(np.asarray(T_ow), d["T_wo_true"])
    return cd, te, re, sr

rows = []
gts = []
for i, d in enumerate(dicts):
    gt = pts(torch.from_numpy(d["z_true"]), np.linalg.inv(d["T_wo_true"]).astype(np.float32))
    gts.append(gt)
    ma = metrics(res["f32"][i].latent, res["f32"][i].T_ow.numpy(), d, gt)
    mb = metrics(res["f16x3"][i].latent, res["f16x3"][i].T_ow.numpy(), d, gt)
    rows.append((ma, mb))
cd_rel = np.array([abs(a[0] - b[0]) / a[0] for a, b in rows])
te_abs = np.array([abs(a[1] - b[1]) for a, b in rows])
re_abs = np.array([abs(a[2] - b[2]) for a, b in rows])
sr_abs = np.array([abs(a[3] - b[3]) for a, b in rows])
print(f"GPU f32 vs GPU f16x3, {B} instances x {iters} iterations ({'pose_known' if pose_known else 'free pose'}):")
print(f"  Chamfer-to-GT [mm]: mean {1e3*np.mean([a[0] for a,_ in rows]):.4f}; relative diff: median {np.median(cd_rel):.2e} p90 {np.percentile(cd_rel,90):.2e} max {cd_rel.max():.2e}")
print(f"  translation-error diff [m]: median {np.median(te_abs):.2e} max {te_abs.max():.2e}; rotation-error diff [deg]: median {np.median(re_abs):.2e} max {re_abs.max():.2e}; scale-ratio diff: median {np.median(sr_abs):.2e} max {sr_abs.max():.2e}")
od = O.fold_decoder(params)
torch.set_num_threads(16)
for i in range(n_oracle):
    d = dicts[i]
    rd = {k: [torch.from_numpy(a) for a in v] for k, v in d["render"].items()}
    t = time.time()
    z, T, n = O.shape_pose_joint_opt(od, cfg, torch.from_numpy(d["latent0"]), torch.from_numpy(d["T_ow0"]), rd, torch.from_numpy(d["points_w"]), d["cube_radius"], pose_known=pose_known)
    # the reference-side noise floor: same oracle, inputs scaled by (1 + 1e-7)
    pw2 = torch.from_numpy(d["points_w"]) * (1 + 1e-7)
    z2, T2, n2 = O.shape_pose_joint_opt(od, cfg, torch.from_numpy(d["latent0"]), torch.from_numpy(d["T_ow0"]), rd, pw2, d["cube_radius"], pose_known=pose_known)
    mo = metrics(z, T.numpy(), d, gts[i]); mo2 = metrics(z2, T2.numpy(), d, gts[i])
    ma, mb = rows[i]
    print(f"inst {i}: iter cpu/gpu {n}/{res['f32'][i].iter_count}  CD[mm] cpu {1e3*mo[0]:.5f} cpu(1e-7 perturbed) {1e3*mo2[0]:.5f} gpu-f32 {1e3*ma[0]:.5f} gpu-f16x3 {1e3*mb[0]:.5f} | rel diff vs cpu: perturbed-cpu {abs(mo2[0]-mo[0])/mo[0]:.2e} gpu-f32 {abs(ma[0]-mo[0])/mo[0]:.2e} gpu-f16x3 {abs(mb[0]-mo[0])/mo[0]:.2e} | t-err[mm] cpu {1e3*mo[1]:.4f} gpu-f32 {1e3*ma[1]:.4f} f16x3 {1e3*mb[1]:.4f} | rot-err[deg] cpu {mo[2]:.4f} f32 {ma[2]:.4f} f16x3 {mb[2]:.4f} | latent rel diff vs cpu: perturbed {float((z2-z).norm()/z.norm()):.2e} f32 {float((res['f32'][i].latent-z).norm()/z.norm()):.2e} f16x3 {float((res['f16x3'][i].latent-z).norm()/z.norm()):.2e}  ({time.time()-t:.0f}s)", flush=True)
