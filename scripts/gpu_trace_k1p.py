"""Per-stage shader-clock attribution of the plain-fp16 decoder kernel (wave 0 of workgroup 0): K loop, wait at the
barrier, epilogue.  Needs the library built with the stamps compiled in:
    python -c "from hortimapping_amd import build; build.build(force=True, extra_flags=['-DHM_K1P_TRACE'])"
GPU box:  python scripts/gpu_trace_k1p.py   (rebuild without the flag afterwards)"""
import ctypes
import sys

sys.path.insert(0, '.')
import torch
from hortimapping_amd import _lib, ops, synthetic as S
from hortimapping_amd.decoder import DecoderWeights

L, B, n = 256, (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 1024
dec = DecoderWeights.from_params(S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)))
dec.set_precision("f16")
lat = 0.05 * torch.randn(B, L, device="cuda")
pts4 = torch.zeros(B, n, 4, device="cuda")
pts4[..., :3] = 0.04 * torch.randn(B, n, 3, device="cuda")
nq = torch.full((B,), n, dtype=torch.int32, device="cuda")
lib = _lib.lib()
lib.hm_debug_set_k1p_trace.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
tr = torch.zeros(80, dtype=torch.int64, device="cuda")
lib.hm_debug_set_k1p_trace(tr.data_ptr())
ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
torch.cuda.synchronize()
lib.hm_debug_set_k1p_trace(None)
t = tr.cpu().numpy().reshape(16, 5)
print("stage   pre-loop     K loop   barrier   epilogue   | stage total (to next stage entry)")
for s in range(16):
    nxt = t[s + 1, 0] if s < 15 else t[s, 4]
    print(f"{s:3d} {t[s,1]-t[s,0]:10d} {t[s,2]-t[s,1]:10d} {t[s,3]-t[s,2]:9d} {t[s,4]-t[s,3]:10d}   | {nxt - t[s,0]:8d}")
print("tile total", t[15, 4] - t[0, 0])
