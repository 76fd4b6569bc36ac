"""Decoders of OTHER layer tables than the shipped 8 x 512 / latent_in = [4] one (deep_sdf_decoder.py:11-72): the
oracle's generalised forward / analytic Jacobian against the g17 fixtures (captured from the reference class by
tests/golden/make_golden_arch.py), and the host-side layer-table logic.  No GPU."""
import numpy as np
import pytest
import torch

from oracle import hm_oracle as O
from tests.golden_util import ARCH_SPECS, load, relmax


def arch_params(name):
    from hortimapping_amd import synthetic as S
    spec = dict(ARCH_SPECS[name])
    return S.make_arch_decoder(**spec)


@pytest.mark.parametrize("name", sorted(ARCH_SPECS))
def test_g17_oracle_matches_the_reference_class(name):
    g = load(f"g17_arch_{name}")
    d = O.fold_decoder(arch_params(name))
    z, x = torch.from_numpy(g["z"]), torch.from_numpy(g["x"])
    assert relmax(O.decoder_forward(d, z, x), g["sdf"]) < 2e-6
    y, jac = O.decoder_jacobian(d, z, x)
    assert relmax(y, g["y"]) < 2e-6
    assert relmax(jac, g["g"]) < 1e-5
    y64, j64 = O.decoder_jacobian(d.to(torch.float64), z, x)          # fp64 tie-breaker
    assert relmax(g["g"], j64) < 1e-5 and relmax(jac, j64) < 1e-5
    assert np.array_equal(g["sdf"], g["y"])                            # decode_sdf and the Jacobian call agree in the reference


def test_g17_analytic_jacobian_equals_autograd_of_the_oracle_forward():
    """Independent of the fixtures: the hand-written backward (LayerNorm, both concatenations, double tanh)."""
    for name in sorted(ARCH_SPECS):
        d = O.fold_decoder(arch_params(name)).to(torch.float64)
        L = d.latent_dim
        gen = torch.Generator().manual_seed(5)
        z = 0.3 * torch.randn(L, generator=gen, dtype=torch.float64)
        x = 0.3 * torch.randn(9, 3, generator=gen, dtype=torch.float64)
        u = torch.cat([z.expand(9, -1), x], 1).requires_grad_(True)
        pre, t, _ = O._layers(d, u, False)
        y = torch.tanh(t if t is not None else pre)
        (ga,) = torch.autograd.grad(y.sum(), u)
        _, g = O.decoder_jacobian(d, z, x)
        assert relmax(g, ga) < 1e-12, name


def test_layer_tables_from_specs_and_from_checkpoints_agree():
    from hortimapping_amd import decoder as D, synthetic as S
    for name, spec in ARCH_SPECS.items():
        p = arch_params(name)
        Ws, bs, ln = D.fold_state_dict_full({k: v for k, v in p.items() if k not in ("latent_dim", "use_tanh")})
        got = D.layer_table(Ws, spec["latent_dim"], ln, spec.get("use_tanh", False))
        ns = {"dims": spec["dims"], "latent_in": spec.get("latent_in", ()), "norm_layers": spec.get("norm_layers", ()),
              "weight_norm": spec.get("weight_norm", False), "xyz_in_all": spec.get("xyz_in_all", False),
              "use_tanh": spec.get("use_tanh", False)}
        want = D.specs_layer_table(spec["latent_dim"], ns)
        assert got == want, name
        assert not D.is_shipped_table(got)
        assert [tuple(w.shape) for w in Ws] == S.arch_layer_dims(spec["latent_dim"], spec["dims"], spec.get("latent_in", ()),
                                                                  spec.get("xyz_in_all", False))
    # the shipped specs.json (deepsdf/models/sweetpepper_32/specs.json:7-18) is recognised as the fast-path table
    shipped = {"dims": [512] * 8, "latent_in": [4], "norm_layers": list(range(8)), "weight_norm": True,
               "xyz_in_all": False, "use_tanh": False}
    for L in (32, 256):
        assert D.is_shipped_table(D.specs_layer_table(L, shipped))
        Ws, _ = S.fold_weight_norm(S.make_synthetic_decoder(L, seed=1))
        assert D.is_shipped_table(D.layer_table(Ws, L))
    assert not D.is_shipped_table(D.specs_layer_table(32, dict(shipped, use_tanh=True)))
    assert not D.is_shipped_table(D.specs_layer_table(32, dict(shipped, weight_norm=False)))      # LayerNorm instead
    with pytest.raises(ValueError, match="layer 0"):
        D.specs_layer_table(32, dict(shipped, latent_in=[0]))
    with pytest.raises(ValueError, match="neither"):
        D.layer_table([np.zeros((64, 35), np.float32), np.zeros((1, 70), np.float32)], 32)


def test_layernorm_parameters_are_folded_with_the_checkpoint():
    from hortimapping_amd.decoder import fold_state_dict_full
    p = arch_params("layernorm")
    sd = {"module." + k: torch.from_numpy(np.asarray(v)) for k, v in p.items() if k not in ("latent_dim", "use_tanh")}
    sd["module.bn3.weight"] = torch.ones(1)          # a bn on the last layer exists in the module; forward never uses it
    sd["module.bn3.bias"] = torch.zeros(1)
    Ws, bs, ln = fold_state_dict_full(sd)
    assert len(Ws) == 4 and sorted(ln) == [0, 1, 2]
    assert np.array_equal(ln[1][0], p["bn1.weight"]) and np.array_equal(ln[1][1], p["bn1.bias"])


def test_create_arch_refuses_bad_layer_tables_without_touching_the_gpu():
    """Argument validation happens before any device call, so it can be checked here (C ABI, no GPU)."""
    import ctypes
    from hortimapping_amd import _lib
    from hortimapping_amd.decoder import HmDecoderArch, MAX_LIN
    lib = _lib.lib()
    w = [np.zeros((64, 35), np.float32), np.zeros((1, 64), np.float32)]
    b = [np.zeros(64, np.float32), np.zeros(1, np.float32)]
    Wp = (_lib.c_float_p * MAX_LIN)(*[a.ctypes.data_as(_lib.c_float_p) for a in w])
    bp = (_lib.c_float_p * MAX_LIN)(*[a.ctypes.data_as(_lib.c_float_p) for a in b])

    def arch(**kw):
        a = HmDecoderArch()
        a.latent_dim, a.n_lin, a.use_tanh = 32, 2, 0
        a.in_dim[0], a.out_dim[0], a.in_dim[1], a.out_dim[1] = 35, 64, 64, 1
        for k, v in kw.items():
            if isinstance(v, tuple):
                getattr(a, k)[v[0]] = v[1]
            else:
                setattr(a, k, v)
        return a
    h = ctypes.c_void_p()
    cases = [(arch(latent_dim=33), "latent_dim"), (arch(n_lin=1), "n_lin"), (arch(n_lin=17), "n_lin"),
             (arch(out_dim=(0, 600)), "widths"), (arch(in_dim=(1, 65)), "does not match"),
             (arch(cat=(0, 1)), "concatenation"), (arch(cat=(1, 1)), "does not match"),
             (arch(layer_norm=(1, 1)), "last layer"), (arch(layer_norm=(0, 1)), "LayerNorm parameters"),
             (arch(out_dim=(1, 2)), "one output")]
    for a, msg in cases:
        rc = lib.hm_decoder_create_arch(ctypes.byref(a), Wp, bp, None, None, ctypes.byref(h))
        assert rc == -1 and msg in lib.hm_last_error().decode(), (msg, lib.hm_last_error())
        assert not h.value


G18_TABLE = dict(latent_dim=32, dims=[128] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=True)   # make_golden_arch.TRAJ_TABLE


def g18_case(name):
    """(decoder params, instance dict in the layout of synthetic.make_instance, fixture) of a g18 trajectory record."""
    from hortimapping_amd import synthetic as S
    g = load(name)
    p = S.make_arch_decoder(seed=21, analytic=True, **G18_TABLE)
    d = {"latent0": g["latent0"], "T_ow0": g["T_ow0"], "points_w": g["points_w"], "cube_radius": float(g["cube_radius"]),
         "render": {k: [g[k]] for k in ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg")}}
    return p, d, g


@pytest.mark.parametrize("name", ["g18_arch_traj_known_5", "g18_arch_traj_known_6", "g18_arch_traj_free_5", "g18_arch_traj_free_6"])
def test_g18_oracle_loop_matches_the_reference_loop_on_another_layer_table(name):
    """Four iterations of the reference's own optimiser on its own Decoder class with a 4 x 128 / latent_in = [2] table."""
    from oracle import hm_oracle as O
    p, d, g = g18_case(name)
    cfg = O.default_opt_cfg()
    cfg["converge"]["max_iter"] = 4
    rd = {k: [torch.from_numpy(a) for a in v] for k, v in d["render"].items()}
    z, T, n = O.shape_pose_joint_opt(O.fold_decoder(p), cfg, torch.from_numpy(d["latent0"].copy()), torch.from_numpy(d["T_ow0"]),
                                     rd, torch.from_numpy(d["points_w"]), d["cube_radius"], pose_known=bool(g["pose_known"]))
    assert n == int(g["iter_count"]) == 4
    # a free pose amplifies rounding differences over the iterations (instance 6: the fp32 oracle is 2e-4 from the fp32 reference
    # in T_ow after four steps); with the pose known the two agree to the last digits
    tz, tT = (1e-3, 2e-5) if bool(g["pose_known"]) else (2e-3, 1e-3)
    assert relmax(z, g["latent"]) < tz and relmax(T, g["T_ow"]) < tT
