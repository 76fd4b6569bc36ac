"""Shader-clock stamps of the staggered f16x3 decoder kernel (k_decoder_hs), workgroup 0, one wave at a time."""
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
tr = torch.zeros(160, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.hm_debug_set_trace.argtypes = [ctypes.c_void_p]
for stag in (1, 0):
    lib.hm_debug_k1h_stagger(stag)
    for wave in (0, 3, 4, 7):
        lib.hm_debug_set_trace_thread(wave * 64)
        for _ in range(2): ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
        tr.zero_()
        lib.hm_debug_set_trace(tr.data_ptr())
        ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
        torch.cuda.synchronize()
        lib.hm_debug_set_trace(None)
        t = tr.cpu().numpy().astype(np.int64)
        if stag:
            print(f"staggered kernel, wave {wave}: per stage: loopA(h0) | wait D | loopA(rest) | loopB | wait B | stores | wait C   total")
            for s in range(16):
                e = t[s*8:(s+1)*8]; nxt = t[(s+1)*8] if s < 15 else t[128]
                if e[1] == 0:   # no pending: stamps 1,2 unset
                    a1, d, a2 = 0, 0, e[3] - e[0]
                else:
                    a1, d, a2 = e[1] - e[0], e[2] - e[1], e[3] - e[2]
                print(f"  s{s:2d}: {a1:7d} {d:6d} {a2:7d} {e[4]-e[3]:7d} {e[5]-e[4]:6d} {e[6]-e[5]:6d} {nxt-e[6]:6d}   {nxt-e[0]:7d}")
            print("  total", t[128] - t[0])
        else:
            print(f"two-barrier kernel, wave {wave}: per stage: loop | wait | epilogue(+top barrier)   total")
            for s in range(16):
                nxt = t[(s+1)*4] if s < 15 else t[64]
                print(f"  s{s:2d}: {t[s*4+1]-t[s*4]:7d} {t[s*4+2]-t[s*4+1]:6d} {nxt-t[s*4+2]:6d}   {nxt-t[s*4]:7d}")
            print("  total", t[64] - t[0])
lib.hm_debug_k1h_stagger(0)
