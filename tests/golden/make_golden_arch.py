#!/usr/bin/env python3
"""Golden vectors for decoders of OTHER layer tables than the shipped one, captured from the ACTUAL reference class.

Run in the build container only (needs the read-only mount /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_arch.py

For every entry of ARCH_SPECS (tests/golden_util.py) the reference's `Decoder(latent_size, dims, ...)`
(deepsdf/networks/deep_sdf_decoder.py:11-72) is instantiated, loaded with the seeded parameters of
`hortimapping_amd.synthetic.make_arch_decoder` (weights are NOT stored: frozen RandomState stream) and evaluated:
  sdf  = wild_completion.utils.decode_sdf(decoder, z, x)                                   (utils.py:144-172)
  y, g = wild_completion.utils.get_batch_sdf_jacobian(decoder, z, x)                       (utils.py:175-193)
`xyz_in_all` tables: get_batch_sdf_jacobian feeds a (n, 1, D) tensor, on which `input[:, -3:]` (deep_sdf_decoder.py:76)
no longer selects xyz, so the reference itself cannot run them through its optimiser; their y / g are taken by the same
autograd call (utils.get_gradient, utils.py:112-122) on the 2-D input that decode_sdf uses, and the fixture says so
(`jac_via` = "autograd_2d").
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

from oracle import ref_shim                      # noqa: E402
from hortimapping_amd import synthetic as S      # noqa: E402
import golden_util as GU                         # noqa: E402


def build(ns, spec):
    kw = {k: v for k, v in spec.items() if k != "seed"}
    params = S.make_arch_decoder(seed=spec["seed"], **kw)
    dec = ns.Decoder(spec["latent_dim"], list(spec["dims"]), dropout=list(range(len(spec["dims"]))), dropout_prob=0.2,
                     norm_layers=list(spec.get("norm_layers", ())), latent_in=list(spec.get("latent_in", ())),
                     weight_norm=spec.get("weight_norm", False), xyz_in_all=spec.get("xyz_in_all", False),
                     use_tanh=spec.get("use_tanh", False), latent_dropout=False)
    sd = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items() if k not in ("latent_dim", "use_tanh")}
    missing = dec.load_state_dict(sd, strict=False)
    # a bn module on the LAST layer exists in __init__ when it is listed in norm_layers, but forward never applies it
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith(f"bn{len(spec['dims'])}.") for k in missing.missing_keys), missing.missing_keys
    dec.eval()
    return dec


def main():
    ns = ref_shim.import_reference()
    rs = np.random.RandomState(777)
    for name, spec in GU.ARCH_SPECS.items():
        L = spec["latent_dim"]
        dec = build(ns, spec)
        z = (0.3 * rs.randn(L)).astype(np.float32)
        x = (0.3 * rs.randn(70, 3)).astype(np.float32)
        zt, xt = torch.from_numpy(z), torch.from_numpy(x)
        sdf = ns.utils.decode_sdf(dec, zt, xt).numpy()
        if spec.get("xyz_in_all", False):
            inp = torch.cat([zt.expand(x.shape[0], -1), xt], 1)
            inp.requires_grad = True
            y = dec(inp)
            g = ns.utils.get_gradient(inp, y)
            y, g, via = y.detach(), g.detach(), "autograd_2d"
        else:
            y, g = ns.utils.get_batch_sdf_jacobian(dec, zt, xt)
            via = "get_batch_sdf_jacobian"
        np.savez_compressed(os.path.join(HERE, f"g17_arch_{name}.npz"), arch=name, z=z, x=x, sdf=sdf,
                            y=y.numpy().reshape(-1), g=g.numpy().reshape(70, L + 3), jac_via=via)
        print(name, "sdf range", float(sdf.min()), float(sdf.max()), "|g| max", float(np.abs(g.numpy()).max()), via)


if __name__ == "__main__":
    main()
