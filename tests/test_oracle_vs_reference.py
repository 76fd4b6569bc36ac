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
