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


TRAJ_TABLE = dict(latent_dim=32, dims=[128] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=True)
TRAJ_SEED = 21


def traj_main():
    """g18: four iterations of the reference's OWN `Optimizer.shape_pose_joint_opt` (optimizer.py:28-302) on its own `Decoder`
    class with a 4 x 128 / latent_in = [2] table (analytic fruit of `make_arch_decoder(analytic=True)`), pose known and free;
    inputs stored explicitly, decoder regenerated from the seed."""
    from oracle import hm_oracle as O
    ns = ref_shim.import_reference()
    p = S.make_arch_decoder(seed=TRAJ_SEED, analytic=True, **TRAJ_TABLE)
    rdec = build(ns, dict(TRAJ_TABLE, seed=TRAJ_SEED, analytic=True))
    od64 = O.fold_decoder(p).to(torch.float64)

    def factory(z):
        zt = torch.from_numpy(np.asarray(z, dtype=np.float64))
        return lambda pts: O.decoder_forward(od64, zt, torch.from_numpy(np.asarray(pts, dtype=np.float64))).numpy()
    cfg = {"device": "cpu", "opt": O.default_opt_cfg(), "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}
    cfg["opt"]["converge"]["max_iter"] = 4
    for pose_known in (True, False):
        for inst_id in (5, 6):
            inst = S.make_instance(None, None, 32, inst_id, n_pts=160, n_frames=1, n_fg=50, n_bg=50, sdf_fn_factory=factory)
            rd = {k: [torch.from_numpy(a) for a in v] for k, v in inst["render"].items()}
            opt = ns.optimizer.Optimizer(cfg, rdec, None, None)
            z, T, n = opt.shape_pose_joint_opt(torch.from_numpy(inst["latent0"].copy()), torch.from_numpy(inst["T_ow0"]), rd,
                                               torch.from_numpy(inst["points_w"]), inst["cube_radius"], None,
                                               pose_known=pose_known)
            name = f"g18_arch_traj_{'known' if pose_known else 'free'}_{inst_id}"
            np.savez_compressed(os.path.join(HERE, name + ".npz"), pose_known=pose_known, latent0=inst["latent0"],
                                T_ow0=inst["T_ow0"], points_w=inst["points_w"], cube_radius=np.float32(inst["cube_radius"]),
                                T_wc=inst["render"]["T_wc"][0], rays_fg=inst["render"]["rays_fg"][0],
                                rays_bg=inst["render"]["rays_bg"][0], depth_fg=inst["render"]["depth_fg"][0],
                                depth_bg=inst["render"]["depth_bg"][0], latent=z.detach().numpy(), T_ow=T.detach().numpy(),
                                iter_count=np.int32(n))
            print(name, "iterations", n, "|z| max", float(z.abs().max()))


if __name__ == "__main__":
    main()
    traj_main()
