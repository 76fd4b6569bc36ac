"""Offline error study of a TWO-unit split arithmetic for the decoder GEMMs (round-3 VERDICT, item 1c):
    W X  ~=  Wh Xh (fp16 MFMA)  +  [Wh | Wl] [Xl ; Xh] as ONE MX-fp8 product (operands rounded to e4m3, per-32 block scales)
against the three-pass f16x3 arithmetic (Wh Xh + Wh Xl + Wl Xh, all fp16 operands, fp32 accumulate) and plain fp32, on
the decoder's own layer shapes: 8 x (512 x 512) layers with ReLU, 4096 queries, weights / activations of the analytic and
of the trained decoder.  CPU emulation (torch float8_e4m3fn rounding, products and sums in fp64 so that only the operand
rounding shows).  Prints the relative error of the layer-8 activations and of the input gradient against fp64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hortimapping_amd import synthetic as S

def f16(x): return x.to(torch.float16).to(torch.float64)
def f8(x, block=32):
    """MX-style e4m3: a power-of-two scale per block of 32 along K, mantissa rounded to 3 bits"""
    shp = x.shape
    xb = x.reshape(-1, block)
    s = torch.exp2(torch.floor(torch.log2(xb.abs().amax(dim=1, keepdim=True).clamp_min(1e-300))) - 7)   # max -> [128, 256) < 448
    q = (xb / s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * s
    return q.reshape(shp)

def layer(W, X, mode):
    W, X = W.double(), X.double()
    if mode == "f64": return W @ X
    if mode == "f32": return (W.float() @ X.float()).double()
    Wh, Xh = f16(W), f16(X)
    Wl, Xl = f16(W - Wh), f16(X - Xh)
    if mode == "f16x3": return Wh @ Xh + Wh @ Xl + Wl @ Xh
    if mode == "f16": return Wh @ Xh
    if mode == "2unit":                      # cross terms with e4m3 operands (K-blocks along the contraction axis)
        return Wh @ Xh + f8(Wh) @ f8(Xl.T).T + f8(Wl) @ f8(Xh.T).T
    raise ValueError(mode)

def run(params, name):
    Ws, bs = S.fold_weight_norm(params)
    L = int(params["latent_dim"])
    g = torch.Generator().manual_seed(0)
    n = 4096
    z = 0.07 * torch.randn(L, generator=g, dtype=torch.float64)
    x = 0.05 * torch.randn(n, 3, generator=g, dtype=torch.float64)
    u = torch.cat([z[None].expand(n, L), x], 1).T                     # (L+3, n)
    out = {}
    for mode in ("f64", "f32", "f16x3", "2unit", "f16"):
        h = u
        for l in range(8):
            W = torch.from_numpy(np.asarray(Ws[l]))
            b = torch.from_numpy(np.asarray(bs[l])).double()[:, None]
            if l == 4: h = torch.cat([h, u], 0)
            h = torch.relu(layer(W, h, mode) + b)
        out[mode] = h
    ref = out["f64"]
    sc = ref.abs().max()
    print(name, {m: f"{float((out[m] - ref).abs().max() / sc):.2e}" for m in ("f32", "f16x3", "2unit", "f16")})

run(S.make_synthetic_decoder(256, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)), "analytic decoder, layer-8 activations, max error / max|h|:")
tr = os.path.join(ROOT, "tests", "golden", "trained_decoder_L256.npz")
with np.load(tr) as f:
    p = {k: (int(f[k]) if k in ("latent_dim", "hidden") else f[k]) for k in f.files}
run(p, "trained decoder,  layer-8 activations, max error / max|h|:")
