"""Handle lifecycle check: create / optimise / destroy many times and watch the device's free memory."""
import gc, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights
p = S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
Ws, bs = S.fold_weight_norm(p)
d = S.make_instance(Ws, bs, 32, 1, n_pts=256, n_frames=2, n_fg=32, n_bg=32)
opt = W.c2_opt_cfg(max_iter=3, n_frame=2)
def free():
    torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0] / 2**20
HO.optimize_batch(DecoderWeights.from_params(p), opt, [W.to_instance(d)] * 4)
f0 = free()
for rnd in range(3):
    for i in range(200):
        dec = DecoderWeights.from_params(p)
        HO.optimize_batch(dec, opt, [W.to_instance(d)] * 4)
        del dec
    print(f"after {200 * (rnd + 1)} cycles: free-memory drift {f0 - free():.1f} MiB", flush=True)
