"""Launch times of the f16x3 decoder kernels over the experiment switches (hm_debug_k1h_variant x hm_debug_k1h_tune):
variant 0 = product kernel k_decoder_h, 1 = ping-pong wave groups, 2 = lockstep + raw barriers + primed four-set ring; tune bits
0-1 priority scheme of the ping-pong kernel, bit 3 four-set weight ring inside k_decoder_h, bit 4 alternating wave priority in its K loop (bit 5: period of twelve steps).  Experimental build only:
HORTIHIP_LIB=hortimapping_amd/variants/libhortihip_exp.so python scripts/gpu_sweep_k1h.py [L] [tunes]"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, ops, _lib
from hortimapping_amd.decoder import DecoderWeights
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tunes = [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 8, 0x10, 0x30]
B, n = 64, 1024
p = S.make_synthetic_decoder(L, seed=5)
dec = DecoderWeights.from_params(p); dec.set_precision('f16x3')
lat = (0.07 * torch.randn(B, L)).float().cuda()
pts4 = torch.zeros(B, n, 4); pts4[..., :3] = 0.04 * torch.randn(B, n, 3); pts4 = pts4.cuda()
nq = torch.full((B,), n, dtype=torch.int32).cuda()
lib = _lib.lib()
EXP = hasattr(lib, 'hm_debug_k1h_variant')          # product library: only the product kernel can be timed

def timed(mode, reps=30):
    for _ in range(3): ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

ref = None
for rnd in range(2):
    for var in (0, 1, 2):
        for tune in tunes:
            if (var != 1 and (tune & 3)) or (var != 0 and (tune & 0x38)): continue
            if not EXP and (var or tune): continue
            if EXP: lib.hm_debug_k1h_variant(var); lib.hm_debug_k1h_tune(tune)
            y, J = ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
            if ref is None: ref = (y.clone(), J.clone())
            same = bool(torch.equal(y, ref[0]) and torch.equal(J, ref[1]))
            print(f"L={L} variant {var} tune {tune:#04x}: fwd+bwd {timed(1):.4f} ms, fwd {timed(0):.4f} ms  bits_equal={same}", flush=True)
if EXP: lib.hm_debug_k1h_variant(0); lib.hm_debug_k1h_tune(0)
