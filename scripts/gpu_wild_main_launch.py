"""wild_pepper-sized batch (64 fruits x 10 frames x 400 rays x 30 samples, L = 32), 3 forced iterations: time per
optimisation with the fused main launch vs the round-2 split sequence (hm_debug_split_render); run under rocprofv3 for the
per-kernel split."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO, _lib
from hortimapping_amd.decoder import DecoderWeights
L = 32
opt = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'wild_pepper.yaml')))['opt']
opt = copy.deepcopy(opt)
opt['converge'].update(max_iter=3, epsilon_g=0.0, epsilon_c=0.0, epsilon_t=0.0, epsilon_r=0.0, epsilon_s=0.0)
p = S.make_synthetic_decoder(L, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(p).set_precision(sys.argv[1] if len(sys.argv) > 1 else 'f16x3')
Ws, bs = S.fold_weight_norm(p)
fac = W.gpu_sdf_factory(dec)
protos = [S.make_instance(Ws, bs, L, i, sdf_fn_factory=fac, n_pts=2000, n_frames=10, n_fg=200, n_bg=200) for i in range(8)]
insts = [W.to_instance(protos[i % 8]) for i in range(64)]
hcfg = HO.opt_cfg_from_dict(opt)
pb = HO.PackedBatch(insts, L, 10, 'cuda')
ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray)
init = (pb.latent.clone(), pb.T_ow.clone())
lib = _lib.lib()
for split in (0, 1, 0, 1):
    lib.hm_debug_split_render(split)
    ts = []
    for rep in range(3):
        pb.latent.copy_(init[0]); pb.T_ow.copy_(init[1])
        torch.cuda.synchronize(); t = time.perf_counter()
        HO.run_packed(ws, hcfg, pb, 0)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"split_render={split}: {min(ts)*1e3:.2f} ms per 3-iteration optimisation of 64 fruits ({min(ts)/3*1e3:.2f} ms / iteration), iters {pb.iter_count.min().item()}..{pb.iter_count.max().item()}")
lib.hm_debug_split_render(0)
