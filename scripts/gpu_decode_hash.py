"""SHA-256 of the f16x3 decode outputs (sdf + Jacobian rows) and of a short joint optimisation on fixed seeded inputs: run it under
two builds of the library (HORTIHIP_LIB=<variant.so>) to check that a kernel change kept the bits."""
import hashlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops, optimizer as HO, workloads as W
from hortimapping_amd.decoder import DecoderWeights
torch.manual_seed(0)
h = hashlib.sha256()
for L in (256, 32):
    for prec in ("f16x3", "f16x3f_f16b"):
        p = S.make_synthetic_decoder(L, seed=5, r0=0.04, aniso=(1.0, 0.75, 1.3))
        dec = DecoderWeights.from_params(p).set_precision(prec)
        g = torch.Generator().manual_seed(L)
        B, n = 7, 320
        lat = (0.07 * torch.randn(B, L, generator=g)).cuda()
        pts4 = torch.zeros(B, n, 4); pts4[..., :3] = 0.04 * torch.randn(B, n, 3, generator=g); pts4 = pts4.cuda()
        nq = torch.tensor([320, 1, 64, 65, 200, 0, 129], dtype=torch.int32).cuda()
        for mode in (0, 1):
            y, J = ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
            h.update(y.cpu().numpy().tobytes())
            if J is not None:
                h.update(J.cpu().numpy().tobytes())
        if L == 32:
            Ws, bs = S.fold_weight_norm(p)
            dec32 = DecoderWeights.from_params(p).set_precision("f32")
            fac = W.gpu_sdf_factory(dec32)
            insts = [W.to_instance(S.make_instance(Ws, bs, L, i, n_pts=300, n_frames=2, n_fg=60, n_bg=40, sdf_fn_factory=fac)) for i in range(18)]
            for r in HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=6, n_sample_on_ray=16, n_frame=2), insts):
                h.update(r.latent.numpy().tobytes()); h.update(r.T_ow.numpy().tobytes()); h.update(bytes([r.iter_count, r.status & 255]))
print("bits", h.hexdigest()[:32])
