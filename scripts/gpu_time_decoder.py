import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops
from hortimapping_amd.decoder import DecoderWeights
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
p = S.make_synthetic_decoder(L, seed=5)
dec = DecoderWeights.from_params(p)
import os
dec.set_precision(os.environ.get('HM_PREC','f32'))
lat = (0.07 * torch.randn(B, L)).float().cuda()
pts4 = torch.zeros(B, n, 4); pts4[..., :3] = 0.04 * torch.randn(B, n, 3)
pts4 = pts4.cuda()
nq = torch.full((B,), n, dtype=torch.int32).cuda()
for mode in (1, 0):
    for _ in range(2):
        ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    t = time.time(); K = 5
    for _ in range(K):
        ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    fl = B * n * (7342080 if mode == 1 else 3671040)
    print(f"mode={mode} L={L} B={B} n={n}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s algorithmic ({fl/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")
