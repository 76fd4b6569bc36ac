"""wild_pepper.yaml as shipped (early exits on), 64 fruits: one optimisation; prints the iteration counts so that the
per-dispatch durations of a rocprofv3 kernel trace can be set against the number of still-active instances."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights
L = 32
opt = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'wild_pepper.yaml')))['opt']
p = S.make_synthetic_decoder(L, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(p).set_precision('f16x3')
Ws, bs = S.fold_weight_norm(p)
fac = W.gpu_sdf_factory(dec)
protos = [S.make_instance(Ws, bs, L, i, sdf_fn_factory=fac, n_pts=2000, n_frames=10, n_fg=200, n_bg=200) for i in range(16)]
insts = [W.to_instance(protos[i % 16]) for i in range(64)]
hcfg = HO.opt_cfg_from_dict(opt)
pb = HO.PackedBatch(insts, L, 10, 'cuda')
ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray)
init = (pb.latent.clone(), pb.T_ow.clone())
for rep in range(2):
    pb.latent.copy_(init[0]); pb.T_ow.copy_(init[1])
    torch.cuda.synchronize(); t = time.perf_counter()
    HO.run_packed(ws, hcfg, pb, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
it = pb.iter_count.cpu().numpy()
active = [int((it > k).sum()) for k in range(int(it.max()))]
print(json.dumps({"ms": dt * 1e3, "active_per_iteration": active}))
