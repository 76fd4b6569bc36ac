"""K1p (plain-fp16 decoder) launch times by HIP events on the launch stream: forward (mode 0) and forward + backward
(mode 1) over B x n queries, L = 256 and L = 32.  GPU box:  [HORTIHIP_LIB=<variant.so>] python scripts/gpu_time_k1p.py [tag]"""
import sys
sys.path.insert(0, '.')
import torch
from hortimapping_amd import synthetic as S, ops
from hortimapping_amd.decoder import DecoderWeights

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
FWD = 3671040
for L in (256, 32):
    dec = DecoderWeights.from_params(S.make_synthetic_decoder(L, seed=5, r0=0.04, aniso=(1.0, 0.75, 1.3))).set_precision("f16")
    B, n = 64, 2048
    lat = (0.07 * torch.randn(B, L)).float().cuda()
    pts4 = torch.zeros(B, n, 4)
    pts4[..., :3] = 0.04 * torch.randn(B, n, 3)
    pts4 = pts4.cuda()
    nq = torch.full((B,), n, dtype=torch.int32).cuda()
    for mode in (0, 1):
        for _ in range(3):
            ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
        torch.cuda.synchronize()
        best, tot, K = 1e9, 0.0, 20
        for _ in range(K):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # the wrapper allocates its outputs first: keep that outside the bracket by timing the second of two calls' kernels
            e0.record()
            ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best, tot = min(best, t), tot + t
        fl = B * n * FWD * (2 if mode == 1 else 1)
        print(f"{tag}: L={L} mode={mode} {B}x{n}: mean {tot / K:.3f} ms  best {best:.3f} ms  -> {fl / (best * 1e-3) / 1e12:.0f} TFLOP/s "
              f"= {fl / (best * 1e-3) / 2.5e15:.3f} of 2.5 PF (dense 8x512 flop, incl. the wrapper's output allocation + memset)", flush=True)
