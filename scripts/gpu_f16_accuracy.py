"""Plain-fp16 decoder (precision f16) against the exact-f32 kernel on the same queries: worst sdf / Jacobian deviation per
latent size, forward-only and forward+backward.  A development aid for changes to hm_decoder_p.hip (tolerance-level
arithmetic: the numbers say whether a change moved the error level).  GPU box:  python scripts/gpu_f16_accuracy.py"""
import sys
sys.path.insert(0, '.')
import torch
from hortimapping_amd import ops, synthetic as S
from hortimapping_amd.decoder import DecoderWeights

for L in (32, 128, 256):
    p = S.make_synthetic_decoder(L, seed=11, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05)
    d16 = DecoderWeights.from_params(p).set_precision("f16")
    d32 = DecoderWeights.from_params(p).set_precision("f32")
    g = torch.Generator().manual_seed(L)
    B, n = 16, 1024
    lat = (0.07 * torch.randn(B, L, generator=g)).cuda()
    pts4 = torch.zeros(B, n, 4)
    pts4[..., :3] = 0.04 * torch.randn(B, n, 3, generator=g)
    pts4 = pts4.cuda()
    nq = torch.full((B,), n, dtype=torch.int32).cuda()
    y0, _ = ops.decode_batch(d16, lat, pts4, nq, mode=0)
    y1, J1 = ops.decode_batch(d16, lat, pts4, nq, mode=1, pose_dim=7)
    yr, Jr = ops.decode_batch(d32, lat, pts4, nq, mode=1, pose_dim=7)
    ey = float((y0 - yr).abs().max())
    ez = float((J1[..., :L] - Jr[..., :L]).abs().max() / Jr[..., :L].abs().max())
    ex = float((J1[..., L:L + 3] - Jr[..., L:L + 3]).abs().max() / Jr[..., L:L + 3].abs().max())
    ep = float((J1[..., L + 3:L + 7] - Jr[..., L + 3:L + 7]).abs().max() / Jr[..., L + 3:L + 7].abs().max())
    print(f"L={L:3d}: fwd==fwd+bwd {bool(torch.equal(y0, y1))}  max|y16-y32| {ey:.2e}  rms {float((y0 - yr).pow(2).mean().sqrt()):.2e}   dz {ez:.2e}  dxyz {ex:.2e}  dpose {ep:.2e}")
