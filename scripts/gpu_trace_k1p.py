"""Per-stage shader-clock attribution of the plain-fp16 decoder kernel (the 8 waves of workgroup 0): K loop, wait at
the barrier, epilogue.  Needs a library built with the stamps compiled in:
    bash scripts/build_variant.sh k1ptrace -DHM_K1P_TRACE
GPU box:  HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_k1ptrace.so python scripts/gpu_trace_k1p.py [B] [mode]"""
import ctypes
import sys

sys.path.insert(0, '.')
import torch
from hortimapping_amd import _lib, ops, synthetic as S
from hortimapping_amd.decoder import DecoderWeights

L, B, n = 256, (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 1024
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dec = DecoderWeights.from_params(S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)))
dec.set_precision("f16")
lat = 0.05 * torch.randn(B, L, device="cuda")
pts4 = torch.zeros(B, n, 4, device="cuda")
pts4[..., :3] = 0.04 * torch.randn(B, n, 3, device="cuda")
nq = torch.full((B,), n, dtype=torch.int32, device="cuda")
lib = _lib.lib()
lib.hm_debug_set_k1p_trace.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
tr = torch.zeros(8 * 80 + 64 + 4, dtype=torch.int64, device="cuda")
lib.hm_debug_set_k1p_trace(tr.data_ptr())
ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
torch.cuda.synchronize()
lib.hm_debug_set_k1p_trace(None)
G = tr.cpu().numpy()[640:704].reshape(8, 8)
C = tr.cpu().numpy()[704:708]
T = tr.cpu().numpy()[:640].reshape(8, 16, 5)
ns = 8 if mode == 0 else 16
t = T[0]
print(f"mode {mode}, wave 0:")
print("stage   pre-loop     K loop   barrier   epilogue   | stage total (to next stage entry)   K-loop end of waves 0..7 relative to the stage's first K-loop start")
for s in range(ns):
    nxt = t[s + 1, 0] if s < ns - 1 else t[s, 4]
    k0 = T[:, s, 1].min()
    ends = " ".join(f"{int(T[w, s, 2] - k0):6d}" for w in range(8))
    starts = " ".join(f"{int(T[w, s, 1] - k0):5d}" for w in range(8))
    print(f"{s:3d} {t[s,1]-t[s,0]:10d} {t[s,2]-t[s,1]:10d} {t[s,3]-t[s,2]:9d} {t[s,4]-t[s,3]:10d}   | {nxt - t[s,0]:8d}   ends {ends}   starts {starts}")
print("tile total", t[ns - 1, 4] - t[0, 0])
if C[2] > C[0]:
    print(f"sustained shader clock between the first and the last stage entry of this launch: {(C[3] - C[1]) / ((C[2] - C[0]) / 100.0):.0f} MHz "
          f"({C[3] - C[1]} shader clocks in {(C[2] - C[0]) / 100.0:.1f} us of the 100 MHz counter; mode 1 = the last launch = backward only)")
if G.any():
    print("stage 1, K-loop group starts (4 steps = 32 MFMAs per group) relative to the first wave's K-loop start:")
    k0 = T[:, 1, 1].min()
    for w in range(8):
        print(f"  wave {w}: start {int(T[w,1,1]-k0):6d}  groups " + " ".join(f"{int(g - k0):6d}" for g in G[w, :7]) + f"  end {int(T[w,1,2]-k0):6d}")
