"""Where the wall time of one optimize_batch call goes at wild_pepper sizes (host packing, workspace, GPU, unpack)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 64
opt = yaml.safe_load(open(os.path.join(ROOT, "configs", "wild_pepper.yaml")))["opt"]
p = S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(p); dec.set_precision("f16x3")
Ws, bs = S.fold_weight_norm(p)
fac = W.gpu_sdf_factory(dec)
protos = [S.make_instance(Ws, bs, 32, i, sdf_fn_factory=fac, n_pts=2000, n_frames=10, n_fg=200, n_bg=200) for i in range(8)]
insts = [W.to_instance(protos[i % 8]) for i in range(B)]
HO.optimize_batch(dec, opt, insts[:8]); torch.cuda.synchronize()
cfg = HO.opt_cfg_from_dict(opt)
for rep in range(2):
    t0 = time.time(); pb = HO.PackedBatch(insts, 32, int(opt["render"]["n_frame"]), torch.device("cuda")); torch.cuda.synchronize()
    t1 = time.time(); ws = HO.Workspace(dec, B, pb.points_stride, pb.F, pb.R, cfg.n_sample_on_ray); torch.cuda.synchronize()
    t2 = time.time(); HO.run_packed(ws, cfg, pb, 0); t2b = time.time(); torch.cuda.synchronize()
    t3 = time.time(); lat = pb.latent.cpu(); T = pb.T_ow.cpu(); it = pb.iter_count.cpu()
    t4 = time.time()
    print(f"pack {1e3*(t1-t0):.0f} ms | workspace ({ws.nbytes/2**20:.0f} MiB) {1e3*(t2-t1):.0f} ms | enqueue {1e3*(t2b-t2):.0f} ms, "
          f"GPU done after {1e3*(t3-t2):.0f} ms | unpack {1e3*(t4-t3):.0f} ms | iterations max {int(it.max())}")
