"""Train a DeepSDF auto-decoder with the reference architecture on a synthetic pepper-like shape family (GPU box, plain
torch; a DATA GENERATOR -- nothing here is the product path or the oracle).

Why: the reference tree ships no decoder weights (`/root/reference/.MISSING_LARGE_BLOBS`), and the analytic decoder of
`hortimapping_amd/synthetic.py` is near-identity in its hidden layers -- the friendliest case for the fp16 range and for
the socket power of the split-operand arithmetic.  This script produces weights with the statistics of a *trained* model
(dense 512 x 512 layers, weight-norm g/v, learnt latent codes) so that parity, the range guard and the bench can be
re-checked on them (`tests/golden/trained_decoder_L256.npz`, `tests/test_gpu_trained_decoder.py`).

Architecture / loss follow the reference's training set-up: 8 weight-normed hidden layers of 512, latent_in [4], ReLU,
final tanh (`deepsdf/networks/deep_sdf_decoder.py:10-110`, `deepsdf/models/sweetpepper_32/specs.json`), clamped L1 on
the SDF + code regulariser, Adam, codes ~ N(0, 1/L) (`deepsdf/train_deep_sdf.py`).  Dropout is left out (it is off at
inference, which is all this repository runs).

Usage (GPU box):  python scripts/train_synthetic_deepsdf.py [--latent 256] [--steps 4000] [--out gpurun_out/trained_decoder_L256.npz]
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

H = 512


class ShapeFamily:
    """Star-shaped 'peppers': r(d) = r0 * ellipsoid(d) * (1 + lobes(d) + taper(d) + bumps(d)), d = p / |p|."""

    def __init__(self, n, seed, device):
        g = torch.Generator().manual_seed(seed)
        u = lambda lo, hi, *s: (lo + (hi - lo) * torch.rand(*s, generator=g)).to(device)
        self.n = n
        self.r0 = u(0.028, 0.042, n)                        # largest radius 0.042 * 1.25 * 1.4 < the 0.08 cube
        self.abc = u(0.80, 1.25, n, 3)
        self.nl = torch.randint(3, 5, (n,), generator=g).to(device).float()
        self.al = u(0.0, 0.10, n)
        self.ph = u(0.0, 2 * math.pi, n)
        self.tp = u(-0.20, 0.20, n)
        self.bd = torch.nn.functional.normalize(torch.randn(n, 4, 3, generator=g), dim=-1).to(device)
        self.ba = u(-0.05, 0.05, n, 4)

    def radius(self, k, d):
        """k: (S,) shape ids, d: (S, Q, 3) unit directions -> (S, Q)."""
        e = torch.rsqrt(((d / self.abc[k][:, None, :]) ** 2).sum(-1))
        phi = torch.atan2(d[..., 1], d[..., 0])
        lob = self.al[k][:, None] * torch.cos(self.nl[k][:, None] * phi + self.ph[k][:, None]) * (1 - d[..., 2] ** 2)
        tap = self.tp[k][:, None] * d[..., 2]
        bump = (self.ba[k][:, None, :] * torch.exp(4.0 * ((d[:, :, None, :] * self.bd[k][:, None, :, :]).sum(-1) - 1))).sum(-1)
        return self.r0[k][:, None] * e * (1 + lob + tap + bump)

    def sdf(self, k, p):
        """First-order signed distance f / |grad f| of f(p) = |p| - r(p / |p|).  p: (S, Q, 3)."""
        p = p.detach().requires_grad_(True)
        with torch.enable_grad():
            n = p.norm(dim=-1).clamp_min(1e-6)
            f = n - self.radius(k, p / n[..., None])
            (g,) = torch.autograd.grad(f.sum(), p)
        return (f / g.norm(dim=-1).clamp_min(0.2)).detach()

    def sample(self, k, q, gen):
        """q queries per shape: 70 % near the surface (sigma 1.5 mm / 5 mm), 30 % uniform in the 0.08 cube."""
        S, dev = k.numel(), k.device
        qn = int(0.7 * q)
        d = torch.nn.functional.normalize(torch.randn(S, qn, 3, generator=gen, device=dev), dim=-1)
        ps = d * self.radius(k, d)[..., None]
        sig = torch.where(torch.rand(S, qn, 1, generator=gen, device=dev) < 0.5, 0.0015, 0.005)
        ps = ps + sig * torch.randn(S, qn, 3, generator=gen, device=dev)
        pu = (torch.rand(S, q - qn, 3, generator=gen, device=dev) * 2 - 1) * 0.08
        p = torch.cat([ps, pu], 1)
        return p, self.sdf(k, p)


def build_decoder(L):
    from hortimapping_amd.synthetic import layer_shapes
    shp = layer_shapes(L, H)                                   # (out, in) of lin0..lin8
    lins = torch.nn.ModuleList(torch.nn.utils.weight_norm(torch.nn.Linear(i, o)) for o, i in shp[:8])
    lins.append(torch.nn.Linear(shp[8][1], shp[8][0]))
    return lins


def forward(lins, z, x):
    u = torch.cat([z, x], -1)
    h = u
    for l in range(9):
        if l == 4:
            h = torch.cat([h, u], -1)
        h = lins[l](h)
        if l < 8:
            h = torch.relu(h)
    return torch.tanh(h[..., 0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=256)
    ap.add_argument("--shapes", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--out", default="gpurun_out/trained_decoder_L256.npz")
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    L, N = a.latent, a.shapes
    fam = ShapeFamily(N, 11, dev)
    lins = build_decoder(L).to(dev)
    codes = torch.nn.Parameter(torch.randn(N, L, device=dev) / math.sqrt(L))
    opt = torch.optim.Adam([{"params": lins.parameters(), "lr": 5e-4}, {"params": [codes], "lr": 1e-3}])
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(1, a.steps // 4), gamma=0.5)
    gen = torch.Generator(device=dev).manual_seed(5)
    S, Q, delta = 64, 2048, 0.1
    t0 = time.time()
    for it in range(a.steps):
        k = torch.randint(0, N, (S,), generator=gen, device=dev)
        p, y = fam.sample(k, Q, gen)
        pred = forward(lins, codes[k][:, None, :].expand(S, Q, L), p)
        loss = (pred.clamp(-delta, delta) - y.clamp(-delta, delta)).abs().mean()
        reg = 1e-4 * min(1.0, it / 100.0) * codes[k].pow(2).sum(1).mean()
        opt.zero_grad(set_to_none=True)
        (loss + reg).backward()
        opt.step()
        sched.step()
        if it % 250 == 0 or it == a.steps - 1:
            print(f"it {it:5d}  L1 {loss.item() * 1e3:.4f} mm  |z| {codes.norm(dim=1).mean().item():.3f}  {time.time() - t0:.0f} s", flush=True)
    # held-out fit + activation range
    with torch.no_grad():
        k = torch.arange(N, device=dev)
        p, y = fam.sample(k, 1024, gen)
        u = torch.cat([codes[k][:, None, :].expand(N, 1024, L), p], -1)
        h, amax = u, []
        for l in range(9):
            if l == 4:
                h = torch.cat([h, u], -1)
            h = lins[l](h)
            amax.append(float(h.abs().max()))
            if l < 8:
                h = torch.relu(h)
        err = (torch.tanh(h[..., 0]) - y).abs()
        near = y.abs() < 0.004
        print(f"held-out |err|: mean {err.mean().item() * 1e3:.4f} mm, near-surface mean {err[near].mean().item() * 1e3:.4f} mm, "
              f"p99 {err.flatten().kthvalue(int(0.99 * err.numel())).values.item() * 1e3:.4f} mm")
        print("max |pre-activation| per layer:", " ".join(f"{v:.2f}" for v in amax))
    out = {"latent_dim": L, "hidden": H, "codes": codes.detach().cpu().numpy().astype(np.float32)}
    for l in range(9):
        m = lins[l]
        if l < 8:
            out[f"lin{l}.weight_v"] = m.weight_v.detach().cpu().numpy().astype(np.float32)
            out[f"lin{l}.weight_g"] = m.weight_g.detach().cpu().numpy().astype(np.float32)
        else:
            out["lin8.weight"] = m.weight.detach().cpu().numpy().astype(np.float32)
        out[f"lin{l}.bias"] = m.bias.detach().cpu().numpy().astype(np.float32)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(a.out, **out)
    print("written", a.out, os.path.getsize(a.out) / 1e6, "MB")


if __name__ == "__main__":
    main()
