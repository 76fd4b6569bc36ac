import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops
from hortimapping_amd.decoder import DecoderWeights
from oracle import hm_oracle as O
torch.manual_seed(0)
for L in (32, 256, 128):
    p = S.make_synthetic_decoder(L, seed=5, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05)
    dec = DecoderWeights.from_params(p)
    import os
    dec.set_precision(os.environ.get('HM_PREC','f32'))
    od = O.fold_decoder(p).to(torch.float64)
    B, n = 3, 200
    nq = [200, 70, 129]
    lat = (0.07 * torch.randn(B, L)).float()
    pts = (0.04 * torch.randn(B, 256, 3)).float()
    pts4 = torch.zeros(B, 256, 4); pts4[..., :3] = pts
    for pose_dim in (0, 7, 6):
        y, J = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), torch.tensor(nq, dtype=torch.int32).cuda(), mode=1, pose_dim=pose_dim)
        y0, _ = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), torch.tensor(nq, dtype=torch.int32).cuda(), mode=0)
        torch.cuda.synchronize()
        y, J, y0 = y.cpu(), J.cpu(), y0.cpu()
        for b in range(B):
            k = nq[b]
            yo, go = O.decoder_jacobian(od, lat[b], pts[b, :k])
            ey = float((y[b, :k] - yo).abs().max() / yo.abs().max())
            ey0 = float((y0[b, :k] - yo).abs().max() / yo.abs().max())
            ez = float((J[b, :k, :L] - go[:, :L]).abs().max() / go[:, :L].abs().max())
            if pose_dim == 0:
                ref = go[:, L:]
            else:
                Jx = O.pose_jacobian(pts[b, :k].double(), pose_dim == 7)
                ref = torch.einsum('ni,nip->np', go[:, L:], Jx)
            ex = float((J[b, :k, L:L + ref.shape[1]] - ref).abs().max() / ref.abs().max())
            pad = float(J[b, k:].abs().max()) if k < 256 else 0.0
            print(f"L={L} pose={pose_dim} b={b} n={k}: y {ey:.2e} y(fwd-only) {ey0:.2e} dz {ez:.2e} dpose {ex:.2e} untouched-pad {pad:.1e}")
