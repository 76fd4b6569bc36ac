"""Build-container-only test: the oracle against the ACTUAL reference imported from /root/reference (skipped on the
GPU box, where the mount does not exist).  Complements the committed golden vectors with fresh random inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference


def test_oracle_matches_live_reference():
    from oracle import hm_oracle as O, ref_shim
    from hortimapping_amd import synthetic as S
    ns = ref_shim.import_reference()
    p = S.make_synthetic_decoder(32, seed=21, aniso=(1.0, 0.8, 1.2), wn_perturb=0.03)
    rdec = ref_shim.build_reference_decoder(ns, p)
    dec = O.fold_decoder(p)
    rs = np.random.RandomState(5)
    z = torch.from_numpy((0.06 * rs.randn(32)).astype(np.float32))
    x = torch.from_numpy((0.04 * rs.randn(50, 3)).astype(np.float32))
    yr, gr = ns.utils.get_batch_sdf_jacobian(rdec, z, x)
    y, g = O.decoder_jacobian(dec, z, x)
    assert float((y - yr.flatten()).abs().max()) < 1e-6 and float((g - gr[:, 0]).abs().max()) < 1e-5
    Ws, bs = S.fold_weight_norm(p)
    inst = S.make_instance(Ws, bs, 32, 3, n_pts=128, n_frames=1, n_fg=60, n_bg=60)
    cfg = {"device": "cpu", "opt": O.default_opt_cfg(), "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}
    cfg["opt"]["converge"]["max_iter"] = 4
    rd = {k: [torch.from_numpy(a) for a in v] for k, v in inst["render"].items()}
    opt = ns.optimizer.Optimizer(cfg, rdec, None, None)
    zr, Tr, nr = opt.shape_pose_joint_opt(torch.from_numpy(inst["latent0"].copy()), torch.from_numpy(inst["T_ow0"]), rd,
                                          torch.from_numpy(inst["points_w"]), 0.08, None, pose_known=True)
    zo, To, no = O.shape_pose_joint_opt(dec, cfg["opt"], torch.from_numpy(inst["latent0"]), torch.from_numpy(inst["T_ow0"]),
                                        rd, torch.from_numpy(inst["points_w"]), 0.08, pose_known=True)
    assert nr == no
    assert float((zo - zr).abs().max() / zr.abs().max()) < 1e-3 and float((To - Tr).abs().max()) < 1e-5


def test_fuzz_slice_oracle_equals_live_reference():
    """A 24-case slice of `scripts/fuzz_oracle_vs_reference.py` (round 5; the 300-case record is
    profiles/r05_oracle_fuzz.txt): random small joint optimisations over every switch of the loop -- Sim(3)/SE(3),
    linear/logistic occupancy, occlusion, the three LM variants, frames without foreground / background rays, frames that
    return None, valid frames that emit zero rays, every exit branch -- the oracle must agree with the imported reference
    on iter_count and exit branch exactly, on the emitted rays per iteration exactly, on H / b to 1e-5 and on the state to
    1e-5 (or the reference's own conditioning noise, measured per case)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_oracle_vs_reference as F
    from oracle import ref_shim
    ns = ref_shim.import_reference()
    bad = []
    for seed in range(1000, 1024):
        ok, rec = F.check_case(ns, seed)
        if not ok:
            bad.append((seed, rec["fails"], rec["iter"], rec["reason"]))
    assert not bad, bad


def _reference_decoder_of(ns, kw, params):
    """The reference's own `Decoder(latent_size, dims, ...)` (deep_sdf_decoder.py:11-72) loaded with `params`."""
    dec = ns.Decoder(kw["latent_dim"], list(kw["dims"]), dropout=list(range(len(kw["dims"]))), dropout_prob=0.2,
                     norm_layers=list(kw.get("norm_layers", ())), latent_in=list(kw.get("latent_in", ())),
                     weight_norm=kw.get("weight_norm", False), xyz_in_all=kw.get("xyz_in_all", False),
                     use_tanh=kw.get("use_tanh", False), latent_dropout=False)
    sd = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items() if k not in ("latent_dim", "use_tanh")}
    missing = dec.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    dec.eval()
    return dec


def test_lm_loops_on_other_layer_tables_oracle_equals_live_reference():
    """Round 5 (decoders of other layer tables): the reference's OWN optimiser run on its own
    `Decoder` class with a table that is not the shipped one -- 4 x 128 with latent_in = [2] (weight norm, analytic fruit)
    through `shape_pose_joint_opt`, and a LayerNorm table through `shape_opt_deepsdf` -- against the oracle's loops on the
    generalised decoder restatement (`oracle/hm_oracle.py: _layers`).  The g17 fixtures pin the decoder functions; this pins
    the loops on top of them."""
    from oracle import hm_oracle as O, ref_shim
    from hortimapping_amd import synthetic as S
    ns = ref_shim.import_reference()
    cfgd = {"device": "cpu", "opt": O.default_opt_cfg(), "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}
    cfgd["opt"]["converge"]["max_iter"] = 4
    # joint loop, skip-connection table
    kw = dict(latent_dim=32, dims=[128] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=True)
    p = S.make_arch_decoder(seed=21, analytic=True, **kw)
    rdec, od = _reference_decoder_of(ns, kw, p), O.fold_decoder(p)
    od64 = od.to(torch.float64)

    def factory(z):
        zt = torch.from_numpy(np.asarray(z, dtype=np.float64))
        return lambda pts: O.decoder_forward(od64, zt, torch.from_numpy(np.asarray(pts, dtype=np.float64))).numpy()
    for pose_known in (True, False):
        inst = S.make_instance(None, None, 32, 5, n_pts=160, n_frames=1, n_fg=50, n_bg=50, sdf_fn_factory=factory)
        rd = {k: [torch.from_numpy(a) for a in v] for k, v in inst["render"].items()}
        opt = ns.optimizer.Optimizer(cfgd, rdec, None, None)
        zr, Tr, nr = opt.shape_pose_joint_opt(torch.from_numpy(inst["latent0"].copy()), torch.from_numpy(inst["T_ow0"]), rd,
                                              torch.from_numpy(inst["points_w"]), 0.08, None, pose_known=pose_known)
        zo, To, no = O.shape_pose_joint_opt(od, cfgd["opt"], torch.from_numpy(inst["latent0"]), torch.from_numpy(inst["T_ow0"]),
                                            rd, torch.from_numpy(inst["points_w"]), 0.08, pose_known=pose_known)
        assert nr == no == 4
        assert float((zo - zr).abs().max() / zr.abs().max()) < 1e-3 and float((To - Tr).abs().max()) < 2e-5, pose_known
    # shape-only loop, LayerNorm table
    kw = dict(latent_dim=32, dims=[128, 160, 128], latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=False)
    p = S.make_arch_decoder(seed=14, **kw)
    rdec, od = _reference_decoder_of(ns, kw, p), O.fold_decoder(p)
    gen = torch.Generator().manual_seed(9)
    pts = 0.05 * torch.randn(300, 3, generator=gen) + torch.tensor([0.0, 0.0, 0.5])
    T_ow = torch.eye(4)
    T_ow[:3, 3] = torch.tensor([0.0, 0.0, -0.5])
    cfgd["opt"]["converge"]["max_iter"] = 3
    opt = ns.optimizer.Optimizer(cfgd, rdec, None, None)
    zr, _, nr = opt.shape_opt_deepsdf(torch.zeros(32), T_ow.clone(), pts, None)
    zo, _, no = O.shape_opt_deepsdf(od, cfgd["opt"], torch.zeros(32), T_ow.clone(), pts)
    assert nr == no == 3
    assert float((zo - zr).abs().max() / zr.abs().max()) < 1e-3


def test_fuzz_slice_layer_tables_oracle_equals_live_reference_class():
    """A 30-table slice of `scripts/fuzz_arch_oracle_vs_reference.py` (the 300-table record is
    profiles/r05_oracle_fuzz_arch.txt): random `dims` / `latent_in` / `xyz_in_all` / weight norm or LayerNorm / `use_tanh`
    tables built by the reference's own `Decoder` class; its `decode_sdf` and input gradients against the oracle's generalised
    restatement (2e-6 / 2e-5; queries sitting on a ReLU kink are identified with the fp64 oracle and not compared)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_arch_oracle_vs_reference as FA
    from oracle import ref_shim
    ns = ref_shim.import_reference()
    bad = []
    for seed in range(7000, 7030):
        ok, kw, errs = FA.check_case(ns, seed)
        if not ok:
            bad.append((seed, kw, errs))
    assert not bad, bad
