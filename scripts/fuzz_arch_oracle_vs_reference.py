#!/usr/bin/env python3
"""The oracle's GENERALISED decoder restatement (oracle/hm_oracle.py: _layers / decoder_forward / decoder_jacobian) against the
LIVE reference `Decoder` class on random layer tables -- build container only (needs /root/reference).

Every case draws a table the reference class can build and run: 1-6 hidden layers of random widths 8 ... 160, a random
`latent_in` subset (never layer 0: the class cannot run that), `xyz_in_all` on / off, `norm_layers` a random subset with or without
`weight_norm` (LayerNorm modules in the second case), `use_tanh` on / off, latent size 8 ... 48; He-scaled random parameters
(hortimapping_amd.synthetic.make_arch_decoder).  Compared on 40 random queries: `decode_sdf` (utils.py:144-172) and the input
gradient -- `get_batch_sdf_jacobian` (utils.py:175-193), or, under `xyz_in_all` where that function cannot run (see
tests/golden/make_golden_arch.py), the same autograd call on the 2-D input -- against the oracle in fp32 (2e-6 / 2e-5 of the largest
magnitude) and, as the tie-breaker, the fp64 oracle.

    python scripts/fuzz_arch_oracle_vs_reference.py [n_cases] [first_seed]     ->  prints a summary, exit code 1 on any disagreement"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import hm_oracle as O, ref_shim                    # noqa: E402
from hortimapping_amd import synthetic as S                     # noqa: E402


def draw_table(seed):
    rs = np.random.RandomState(seed)
    n_hidden = int(rs.randint(1, 7))
    L = int(rs.choice([8, 16, 24, 32, 48]))
    latent_in = sorted(int(i) for i in rs.choice(np.arange(1, n_hidden + 1), size=int(rs.randint(0, min(3, n_hidden) + 1)), replace=False))
    xyz_in_all = bool(rs.rand() < 0.35)
    # every layer's own width must leave room for what is concatenated in front of the next one (deep_sdf_decoder.py:41-47)
    dims = [int(rs.randint(L + 3 + 8, 161)) for _ in range(n_hidden)]
    weight_norm = bool(rs.rand() < 0.5)
    norm_layers = sorted(int(i) for i in np.flatnonzero(rs.rand(n_hidden + 1) < 0.6))
    return dict(latent_dim=L, dims=dims, latent_in=latent_in, norm_layers=norm_layers, weight_norm=weight_norm,
                xyz_in_all=xyz_in_all, use_tanh=bool(rs.rand() < 0.3))


def reference_decoder(ns, kw, params):
    dec = ns.Decoder(kw["latent_dim"], list(kw["dims"]), dropout=list(range(len(kw["dims"]))), dropout_prob=0.2,
                     norm_layers=list(kw["norm_layers"]), latent_in=list(kw["latent_in"]), weight_norm=kw["weight_norm"],
                     xyz_in_all=kw["xyz_in_all"], use_tanh=kw["use_tanh"], latent_dropout=False)
    sd = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items() if k not in ("latent_dim", "use_tanh")}
    missing = dec.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    n_lin = len(kw["dims"]) + 1
    assert all(k.startswith(f"bn{n_lin - 1}.") for k in missing.missing_keys), missing.missing_keys   # unused bn of the last layer
    dec.eval()
    return dec


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_case(ns, seed):
    kw = draw_table(seed)
    p = S.make_arch_decoder(seed=seed, **kw)
    rdec, od = reference_decoder(ns, kw, p), O.fold_decoder(p)
    L = kw["latent_dim"]
    rs = np.random.RandomState(seed + 100000)
    z = torch.from_numpy((0.3 * rs.randn(L)).astype(np.float32))
    x = torch.from_numpy((0.3 * rs.randn(40, 3)).astype(np.float32))
    sdf = ns.utils.decode_sdf(rdec, z, x).numpy()
    if kw["xyz_in_all"]:
        inp = torch.cat([z.expand(40, -1), x], 1)
        inp.requires_grad = True
        y = rdec(inp)
        g = ns.utils.get_gradient(inp, y).detach().numpy()
        y = y.detach().numpy().reshape(-1)
    else:
        y, g = ns.utils.get_batch_sdf_jacobian(rdec, z, x)
        y, g = y.numpy().reshape(-1), g.numpy().reshape(40, L + 3)
    yo, go = O.decoder_jacobian(od, z, x)
    od64 = od.to(torch.float64)
    y64, g64 = O.decoder_jacobian(od64, z, x)
    # A hidden unit whose pre-activation is within fp32 rounding of 0 sits ON its ReLU kink: there the gradient jumps by a finite
    # amount between ANY two arithmetics (the reference's addmm and the oracle's matmul + add round differently).  Such queries are
    # found by comparing the fp32 oracle's ReLU masks with the fp64 oracle's and left out of the gradient comparison (counted).
    u32 = O._inputs(od, z, x)
    m32 = O._layers(od, u32, True)[2]
    m64 = O._layers(od64, O._inputs(od64, z, x), True)[2]
    kink = np.zeros(40, dtype=bool)
    for a, b in zip(m32, m64):
        kink |= (a[0] != b[0]).any(dim=1).numpy()
    keep = ~kink
    errs = {"sdf": rel(O.decoder_forward(od, z, x).numpy(), sdf), "y": rel(yo.numpy(), y), "g": rel(go.numpy()[keep], g[keep]),
            "g_ref_vs_fp64": rel(g[keep], g64.numpy()[keep]), "g_oracle_vs_fp64": rel(go.numpy()[keep], g64.numpy()[keep]),
            "kink_queries": int(kink.sum())}
    ok = errs["sdf"] < 2e-6 and errs["y"] < 2e-6 and errs["g"] < 2e-5 and errs["g_oracle_vs_fp64"] < 2e-5 and kink.sum() <= 2
    return ok, kw, errs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ns = ref_shim.import_reference()
    bad, worst = [], {}
    stats = {"latent_in": 0, "xyz_in_all": 0, "layernorm": 0, "weight_norm": 0, "use_tanh": 0}
    for seed in range(first, first + n):
        ok, kw, errs = check_case(ns, seed)
        for k, v in errs.items():
            worst[k] = (worst.get(k, 0) + v) if k == "kink_queries" else max(worst.get(k, 0.0), v)
        stats["latent_in"] += bool(kw["latent_in"]); stats["xyz_in_all"] += kw["xyz_in_all"]; stats["use_tanh"] += kw["use_tanh"]
        stats["layernorm"] += (not kw["weight_norm"]) and any(l < len(kw["dims"]) for l in kw["norm_layers"])
        stats["weight_norm"] += kw["weight_norm"] and bool(kw["norm_layers"])
        if not ok:
            bad.append((seed, kw, errs))
    print(f"{n} random layer tables (seeds {first} ... {first + n - 1}): {len(bad)} disagreements")
    print("tables with: " + ", ".join(f"{k} {v}" for k, v in stats.items()))
    print("worst relative differences: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items() if k != "kink_queries") +
          f"; queries on a ReLU kink (fp32 and fp64 masks differ; gradient not compared): {worst.get('kink_queries', 0)} of {40 * n}")
    for b in bad[:10]:
        print("DISAGREE", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
