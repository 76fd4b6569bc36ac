"""Layer-by-layer check of the plain-fp16 decoder kernel: X[row][query] as stored by forward stage s (debug build,
-DHM_K1P_DEBUG) against a numpy forward pass.   bash scripts/build_variant.sh k1pdbg -DHM_K1P_DEBUG
GPU box: HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_k1pdbg.so python scripts/gpu_probe_k1p.py [L]"""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import _lib, ops, synthetic as S
from hortimapping_amd.decoder import DecoderWeights

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = S.make_synthetic_decoder(L, seed=11, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05)
Ws, bs = S.fold_weight_norm(p)
dec = DecoderWeights.from_params(p).set_precision("f16")
g = torch.Generator().manual_seed(L)
B, n = 2, 256
lat = 0.07 * torch.randn(B, L, generator=g)
pts = 0.04 * torch.randn(B, n, 3, generator=g)
pts4 = torch.zeros(B, n, 4); pts4[..., :3] = pts
nq = torch.full((B,), n, dtype=torch.int32)
lib = _lib.lib()
lib.hm_debug_set_k1p_probe.argtypes = [ctypes.c_int, ctypes.c_int]
# numpy forward, instance 0
z = lat[0].numpy().astype(np.float64); X = pts[0].numpy().astype(np.float64)
m = 509 - L
acts = []
x0 = np.concatenate([np.repeat(z[None], n, 0), X], 1)
h = np.maximum(x0 @ np.asarray(Ws[0], np.float64).T + bs[0], 0); acts.append(h)
for l in (1, 2):
    h = np.maximum(h @ np.asarray(Ws[l], np.float64).T + bs[l], 0); acts.append(h)
h3 = np.maximum(h @ np.asarray(Ws[3], np.float64).T + bs[3], 0)
x3 = np.zeros((n, 512)); x3[:, :m] = h3; x3[:, m:m + 3] = X; acts.append(x3)
x4 = np.concatenate([h3, x0], 1)
h = np.maximum(x4 @ np.asarray(Ws[4], np.float64).T + bs[4], 0); acts.append(h)
for l in (5, 6, 7):
    h = np.maximum(h @ np.asarray(Ws[l], np.float64).T + bs[l], 0); acts.append(h)
for s in range(8):
    worst = 0.0; line = []
    for row in (0, 1, 3, 4, 7, 8, 12, 15, 16, 31, 32, 100, 250, 253, 255, 256, 300, 511):
        if s == 3 and row >= 512 - L: continue
        lib.hm_debug_set_k1p_probe(s, row)
        y, _ = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), nq.cuda(), mode=0)
        got = y[0].cpu().numpy().astype(np.float64)
        ref = acts[s][:, row]
        err = float(np.abs(got - ref).max()); sc = float(np.abs(acts[s]).max())
        line.append(f"{row}:{err / sc:.1e}")
    print(f"stage {s} (max|act| {np.abs(acts[s]).max():.3f}) rel err by row  " + " ".join(line))
lib.hm_debug_set_k1p_probe(-1, 0)
