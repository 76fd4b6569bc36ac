"""Shader-clock stamps + launch times of the f16x3 decoder kernels: k_decoder_h (product, variant 0), k_decoder_g ping-pong
(variant 1) and lockstep (variant 2), workgroup 0, one wave at a time.  Experimental build only (HORTIHIP_LIB=...exp.so)."""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops, _lib
from hortimapping_amd.decoder import DecoderWeights
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, n = 64, 1024
p = S.make_synthetic_decoder(L, seed=5)
dec = DecoderWeights.from_params(p); dec.set_precision('f16x3')
lat = (0.07 * torch.randn(B, L)).float().cuda()
pts4 = torch.zeros(B, n, 4); pts4[..., :3] = 0.04 * torch.randn(B, n, 3); pts4 = pts4.cuda()
nq = torch.full((B,), n, dtype=torch.int32).cuda()
tr = torch.zeros(160, dtype=torch.int64, device='cuda')
lib = _lib.lib()
lib.hm_debug_set_trace.argtypes = [ctypes.c_void_p]

def timed(mode, reps=20):
    for _ in range(3): ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for rnd in range(2):
    for var in (0, 1, 2):
        lib.hm_debug_k1h_variant(var)
        print(f"L={L} variant {var} ({('two-barrier (product)', 'ping-pong', 'lockstep + primed ring')[var]}): fwd+bwd {timed(1):.4f} ms, fwd {timed(0):.4f} ms  ({B} x {n} queries)")
for var in (1, 2, 0):
    lib.hm_debug_k1h_variant(var)
    for wave in (0, 3, 4, 7):
        lib.hm_debug_set_trace_thread(wave * 64)
        for _ in range(2): ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
        tr.zero_()
        lib.hm_debug_set_trace(tr.data_ptr())
        ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
        torch.cuda.synchronize()
        lib.hm_debug_set_trace(None)
        t = tr.cpu().numpy().astype(np.int64)
        if var == 2:  # lockstep
            print(f"lockstep kernel with primed ring, wave {wave}: per stage: side+loop | wait | E+prime | wait   total")
            for s in range(16):
                e = t[s*8:(s+1)*8]; nxt = t[(s+1)*8] if s < 15 else t[128]
                print(f"  s{s:2d}: {e[3]-e[0]:7d} {e[4]-e[3]:6d} {e[5]-e[4]:6d} {e[6]-e[5]:6d} (+{nxt-e[6]})   {nxt-e[0]:7d}")
            print("  total", t[128] - t[0])
        elif var == 1:
            print(f"ping-pong kernel, wave {wave}: per stage: P1 | wait | P2 | wait | E | wait   total")
            for s in range(16):
                e = t[s*8:(s+1)*8]; nxt = t[(s+1)*8] if s < 15 else t[128]
                print(f"  s{s:2d}: {e[1]-e[0]:7d} {e[2]-e[1]:6d} {e[3]-e[2]:7d} {e[4]-e[3]:6d} {e[5]-e[4]:6d} {e[6]-e[5]:6d} (+{nxt-e[6]})   {nxt-e[0]:7d}")
            print("  total", t[128] - t[0])
        else:
            print(f"two-barrier kernel, wave {wave}: per stage: loop | wait | epilogue(+top barrier)   total")
            for s in range(16):
                nxt = t[(s+1)*4] if s < 15 else t[64]
                print(f"  s{s:2d}: {t[s*4+1]-t[s*4]:7d} {t[s*4+2]-t[s*4+1]:6d} {nxt-t[s*4+2]:6d}   {nxt-t[s*4]:7d}")
            print("  total", t[64] - t[0])
lib.hm_debug_k1h_variant(0)
