#!/usr/bin/env python3
"""GPU box: how far is the ONE-pass fp16 decoder forward (K1p, 128-query tiles) from the f16x3 forward on the samples the
render term decodes?  This difference is the margin `eps` of the linear-occupancy screening pass (round 5,
hm_optimize.hip): a ray sample whose fp16 sdf lies beyond occ_cutoff + eps is taken as far from the band, so eps must
bound |s_f16 - s_f16x3| with room to spare.

    python scripts/measure_screen_eps.py > gpurun_out/screen_eps.txt

Decoders: the analytic L = 32 pepper / berry of the shipped-configuration benches, the analytic L = 256 bench decoder and
the TRAINED L = 256 decoder (dense layers); latents: zero (the start), N(0, 0.07^2) draws (the synthetic fruits' range)
and 3 x that; points: uniform in the ball of radius 1.5 x the fruit's cube radius (scale changes move the ball)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hortimapping_amd import ops, synthetic as S          # noqa: E402
from hortimapping_amd.decoder import DecoderWeights       # noqa: E402


def trained():
    with np.load(os.path.join(ROOT, "tests", "golden", "trained_decoder_L256.npz")) as f:
        return {k: (int(f[k]) if k in ("latent_dim", "hidden") else f[k]) for k in f.files}


def main():
    torch.cuda.set_device(0)
    cases = [("pepper32", S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3)), 0.08),
             ("berry32", S.make_synthetic_decoder(32, seed=1, r0=0.02, aniso=(1.0, 0.75, 1.3)), 0.04),
             ("pepper256", S.make_synthetic_decoder(256, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)), 0.08),
             ("trained256", trained(), 0.08)]
    g = torch.Generator().manual_seed(7)
    print("decoder      latents        n        max|d|     p99.99     p99        max|d| where |s|<0.02   max|s|")
    worst = 0.0
    for name, p, rad in cases:
        L = int(p["latent_dim"])
        dh = DecoderWeights.from_params(p).set_precision("f16x3")
        dp = DecoderWeights.from_params(p).set_precision("f16")
        d32 = DecoderWeights.from_params(p).set_precision("f32")
        B, N = 32, 16384
        codes = p.get("codes")
        for lname, sig in (("zero", 0.0), ("sigma0.07", 0.07), ("sigma0.21", 0.21)) + ((("codes", -1.0),) if codes is not None else ()):
            if sig < 0:
                idx = torch.randint(0, codes.shape[0], (B,), generator=g)
                lat = torch.from_numpy(np.asarray(codes, dtype=np.float32))[idx]
            else:
                lat = sig * torch.randn(B, L, generator=g)
            d = torch.randn(B, N, 3, generator=g)
            d = d / d.norm(dim=-1, keepdim=True) * (1.5 * rad * torch.rand(B, N, 1, generator=g) ** (1 / 3))
            pts4 = torch.zeros(B, N, 4)
            pts4[..., :3] = d
            nq = torch.full((B,), N, dtype=torch.int32)
            a = [t.cuda().contiguous() for t in (lat, pts4, nq)]
            yh, _ = ops.decode_batch(dh, *a, mode=0)
            yp, _ = ops.decode_batch(dp, *a, mode=0)
            y32, _ = ops.decode_batch(d32, *a, mode=0)
            diff = (yp - yh).abs().flatten().double().cpu()
            near = (yh.abs() < 0.02).flatten().cpu()
            q = torch.quantile(diff[::7], torch.tensor([0.9999, 0.99], dtype=torch.float64))
            dn = float(diff[near].max()) if near.any() else 0.0
            worst = max(worst, float(diff.max()))
            print(f"{name:12s} {lname:10s} {diff.numel():9d}   {float(diff.max()):.3e}  {float(q[0]):.3e}  {float(q[1]):.3e}  "
                  f"{dn:.3e}               {float(yh.abs().max()):.3f}   (f16x3 vs f32: {float((yh - y32).abs().max()):.2e})")
    print(f"worst |s_f16 - s_f16x3| over everything: {worst:.3e}")


if __name__ == "__main__":
    main()
