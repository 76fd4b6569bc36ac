import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops, _lib
from hortimapping_amd.decoder import DecoderWeights
L, B, n = 256, 64, 1024
p = S.make_synthetic_decoder(L, seed=5)
dec = DecoderWeights.from_params(p); dec.set_precision('f16x3')
lat = (0.07 * torch.randn(B, L)).float().cuda()
pts4 = torch.zeros(B, n, 4); pts4[..., :3] = 0.04 * torch.randn(B, n, 3); pts4 = pts4.cuda()
nq = torch.full((B,), n, dtype=torch.int32).cuda()
tr = torch.zeros(80, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.hm_debug_set_trace.argtypes = [ctypes.c_void_p]
for mode in (1, 0):
    for _ in range(2): ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    lib.hm_debug_set_trace(tr.data_ptr())
    ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    lib.hm_debug_set_trace(None)
    t = tr.cpu().numpy().astype(np.int64)
    ns = 16 if mode == 1 else 8
    print(f"mode {mode}: stage: hook+gemm, barrier wait, epilogue(+next top barrier)   [shader clock ticks]")
    tot_g = tot_b = tot_e = 0
    for s in range(ns):
        g = t[s*4+1] - t[s*4+0]; bw = t[s*4+2] - t[s*4+1]
        nxt = t[(s+1)*4+0] if s + 1 < ns else t[ns*4]
        e = nxt - t[s*4+2]
        tot_g += g; tot_b += bw; tot_e += e
        print(f"  s{s:2d}: {g:8d} {bw:8d} {e:8d}")
    print(f"  total gemm {tot_g} barrier {tot_b} epilogue {tot_e} sum {tot_g+tot_b+tot_e}")
