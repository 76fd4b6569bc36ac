"""The 200-iteration, L = 256 gate pinned on the REFERENCE ITSELF (CPU tier).

`tests/golden/c2_fullsize_reference.npz` holds the results of the actual reference loop
(`/root/reference/wild_completion/optimizer.py:28-302`, run by `tests/golden/make_reference_records.py` through
`oracle/ref_shim.py`: its autograd Jacobians, its torch.inverse, its loss builders) on instances of the C2 full-size
fixture, both pose modes, nominal inputs + the four structured 1e-7 perturbations.  `c2_fullsize_oracle.npz` holds the
oracle's results on the same inputs and perturbations.  This file checks, without a GPU,
  (a) the oracle's nominal result deviates from the reference's nominal result like one more perturbed reference run
      (rank test over instances x modes, tests/parity_stats.py), in state AND in the parity metrics;
  (b) the reference's own perturbation noise and the oracle's agree within 2x (geometric mean over instances) -- so the
      noise that calibrates the GPU gate of tests/test_gpu_fullsize.py is the reference's, not an artefact of the oracle.
The metrics on the CPU use the fp32 oracle decoder as the (single) sampler with fewer directions than the GPU tier."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
import parity_stats as PS

L = 256
N_CD = 3                      # instances whose Chamfer distance is sampled on the CPU (the decoder forward is the cost)


def _load(prefix="c2_fullsize"):
    ref = np.load(os.path.join(GOLDEN_DIR, prefix + "_reference.npz"))
    orc = np.load(os.path.join(GOLDEN_DIR, prefix + "_oracle.npz"))
    inp = np.load(os.path.join(GOLDEN_DIR, prefix + "_inputs.npz"))
    ids = ref["inst_ids"]
    assert list(ref["perts"]) == list(orc["perts"][:len(ref["perts"])])      # same perturbations, same order
    return ref, orc, inp, ids


def test_reference_records_are_full_length_runs():
    ref, orc, inp, ids = _load()
    assert len(ids) >= 8 and int(ref["n_iter"]) == 200
    for m in ("known", "free"):
        assert np.all(ref[f"{m}_iter_count"] == 200) and np.isfinite(ref[f"{m}_latent"]).all()
        # pose_known runs must not move the pose except for the scale (optimizer.py:237-238)
    R0 = inp["T_ow0"][ids][:, :3, :3]
    Rk = ref["known_T_ow"][0][:, :3, :3]
    s = np.cbrt(np.linalg.det(Rk) / np.linalg.det(R0))
    assert np.abs(Rk / s[:, None, None] - R0).max() < 1e-5


def _state_dev(a, b):
    """per-instance max |difference| of latent and of T_ow"""
    return np.abs(a[0] - b[0]).reshape(a[0].shape[0], -1).max(axis=1), np.abs(a[1] - b[1]).reshape(a[1].shape[0], -1).max(axis=1)


@pytest.mark.parametrize("prefix", ["c2_fullsize", "trained_c2"])
def test_oracle_deviates_from_reference_like_one_more_perturbed_run_state_level(prefix):
    """(a) at state level, all instances x both modes: rank of |oracle_nominal - reference_nominal| among the K
    |reference_pert - reference_nominal| (K = 4 on the analytic decoder, 16 instances; K = 2 on the TRAINED decoder with
    its dense, kinked layers, 6 instances); and (b) noise levels within 2x (analytic) / 3x (trained, 2 draws each)."""
    ref, orc, inp, ids = _load(prefix)
    us, ratios = [], []
    for m in ("known", "free"):
        zr, Tr = ref[f"{m}_latent"], ref[f"{m}_T_ow"]
        zo, To = orc[f"{m}_latent"][:, ids], orc[f"{m}_T_ow"][:, ids]
        K = zr.shape[0] - 1
        for r_all, o_all in ((zr, zo), (Tr, To)):
            n = r_all.shape[1]
            flat = lambda x: x.reshape(x.shape[0], n, -1)
            dev = np.abs(flat(o_all)[0] - flat(r_all)[0]).max(axis=1)
            pert_r = np.abs(flat(r_all)[1:] - flat(r_all)[:1]).max(axis=2)                  # (K, n) reference noise
            pert_o = np.abs(flat(o_all)[1:K + 1] - flat(o_all)[:1]).max(axis=2)              # (K, n) oracle noise, same perts
            us.append(PS.rank_fraction(dev, pert_r))
            ratios.append(np.log(pert_o.max(axis=0) / pert_r.max(axis=0)))
            # (no per-instance cap: the largest of FOUR heavy-tailed draws is too coarse a yardstick for a fifth)
    u = np.concatenate(us)
    K = ref["known_latent"].shape[0] - 1
    dplus, p = PS.ks_upper(u, K)
    print(f"\n{prefix}: state-level ranks of |oracle - reference| among the reference's {K} perturbed runs: n={len(u)} mean "
          f"{u.mean():.2f} KS+ {dplus:.3f} p {p:.3f}")
    assert p >= 1e-3 and u.mean() < 0.8
    g = float(np.exp(np.mean(np.concatenate(ratios))))
    print(f"{prefix}: oracle noise / reference noise (geometric mean over instances, modes, latent and pose): {g:.2f}")
    lim = 2.0 if K >= 4 else 3.0
    assert 1.0 / lim <= g <= lim


def _cpu_metrics(od, lat, T_ow, gt_world, T_wo_true, dirs):
    """(Chamfer-to-GT, translation error, rotation error, scale ratio) with the fp32 oracle decoder as the sampler."""
    from hortimapping_amd import metrics as MX
    from oracle import hm_oracle as O
    z = torch.from_numpy(np.asarray(lat, dtype=np.float32))

    def sdf(p):
        return O.decoder_forward(od, z, torch.from_numpy(np.asarray(p, dtype=np.float32))).numpy().reshape(-1)
    lo, hi = np.zeros(len(dirs)), np.full(len(dirs), 0.08)
    for _ in range(18):
        mid = 0.5 * (lo + hi)
        ins = sdf(dirs * mid[:, None]) < 0
        lo, hi = np.where(ins, mid, lo), np.where(ins, hi, mid)
    p_o = dirs * (0.5 * (lo + hi))[:, None]
    if gt_world is None:
        return p_o
    T_wo = np.linalg.inv(np.asarray(T_ow, dtype=np.float64))
    pw = p_o @ T_wo[:3, :3].T + T_wo[:3, 3]
    return np.array([MX.chamfer_distance(pw, gt_world), *MX.pose_error(np.asarray(T_ow), T_wo_true)])


def test_oracle_deviates_from_reference_like_one_more_perturbed_run_metric_level():
    """(a) in the metrics BASELINE.json names (Chamfer-to-ground-truth, pose error), on N_CD instances x both modes:
    |m_oracle - m_reference| against the reference's own perturbation band, and (b) the bands within 2x."""
    from hortimapping_amd import metrics as MX, synthetic as S
    from oracle import hm_oracle as O
    ref, orc, inp, ids = _load()
    od = O.fold_decoder(S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)))
    dirs = MX.fibonacci_dirs(300)
    us, ratios = [], []
    for k, i in enumerate(ids[:N_CD]):
        Ttrue = inp["T_wo_true"][i].astype(np.float64)
        gt = _cpu_metrics(od, inp["z_true"][i], None, None, None, dirs) @ Ttrue[:3, :3].T + Ttrue[:3, 3]
        for m in ("known", "free"):
            mr = np.stack([_cpu_metrics(od, ref[f"{m}_latent"][p, k], ref[f"{m}_T_ow"][p, k], gt, Ttrue, dirs) for p in range(5)])
            mo = np.stack([_cpu_metrics(od, orc[f"{m}_latent"][p, i], orc[f"{m}_T_ow"][p, i], gt, Ttrue, dirs) for p in range(5)])
            dev = np.abs(mo[0] - mr[0])
            pert_r = np.abs(mr[1:] - mr[0])
            pert_o = np.abs(mo[1:] - mo[0])
            scale = np.array([mr[0, 0], max(mr[0, 1], 1e-3), max(mr[0, 2], 0.1), 1.0])
            floor = 1e-4 * scale
            print(f"inst {i} {m}: rel CD |oracle-ref| {dev[0] / mr[0, 0]:.2e}, reference noise {pert_r[:, 0].max() / mr[0, 0]:.2e}, "
                  f"oracle noise {pert_o[:, 0].max() / mr[0, 0]:.2e}")
            for c in range(4):
                if dev[c] > floor[c]:
                    us.append(PS.rank_fraction(dev[c:c + 1], pert_r[:, c:c + 1])[0])
                if pert_r[:, c].max() > floor[c] and pert_o[:, c].max() > floor[c]:
                    ratios.append(np.log(pert_o[:, c].max() / pert_r[:, c].max()))
    u = np.array(us)
    dplus, p = PS.ks_upper(u, 4)
    print(f"metric-level ranks: n={len(u)} mean {u.mean() if len(u) else 0.5:.2f} KS+ {dplus:.3f} p {p:.3f}")
    assert p >= 1e-3
    if ratios:
        g = float(np.exp(np.mean(ratios)))
        print(f"oracle / reference metric noise (geometric mean): {g:.2f}")
        assert 0.5 <= g <= 2.0


def test_rank_statistics_helper():
    rs = np.random.RandomState(0)
    D = np.abs(rs.standard_cauchy((16, 400)))
    d = np.abs(rs.standard_cauchy(400))
    g = PS.gate(d, D, np.zeros(400))
    assert g["ok"] and abs(g["mean_rank"] - 0.5) < 0.06                    # same law: uniform ranks
    g2 = PS.gate(3.0 * d, D, np.zeros(400))
    assert not g2["ok"] and g2["mean_rank"] > 0.6                            # 3x larger deviations are detected
    g3 = PS.gate(d, D, np.full(400, 1e9))
    assert g3["ok"] and g3["outright"] == 400 and g3["ranked"] == 0


def test_candidate_side_selection_biases_the_rank_gate_and_symmetric_selection_does_not():
    """Round 5.  Exchangeable synthetic data -- candidate and 16 stand-ins drawn from ONE law per
    instance -- with a floor that removes about two thirds of the instances, as the 1e-4 floor does for the scale error of
    the C2-joint gate.  Ranking only the instances whose CANDIDATE exceeds the floor (rounds 3-4) tilts the mean rank to
    ~0.7 and rejects a true null far more often than alpha; the symmetric rule (largest of all 17 deviations above the
    floor) leaves the ranks uniform.  This is the attribution of the GPU's 'high-side rank bias': a property of the gate,
    shared by every one-operation variant of the oracle itself (profiles/r05_rank_bias_table.txt)."""
    rs = np.random.RandomState(1)
    n, K, trials = 64, 16, 400
    rej = {"candidate": 0, "symmetric": 0}
    mean_rank = {"candidate": [], "symmetric": []}
    for _ in range(trials):
        scale = np.exp(rs.normal(0.0, 1.5, n))                       # instance-to-instance spread of the noise level
        allv = np.abs(rs.standard_normal((K + 1, n))) * scale
        floor = np.full(n, np.quantile(allv[0], 0.66))                # ~2/3 of the candidates inside the floor
        for sel in rej:
            g = PS.gate(allv[0], allv[1:], floor, alpha=0.01, selection=sel)
            rej[sel] += (not g["ok"]) and g["has_power"]
            if g["ranked"]:
                mean_rank[sel].append(g["mean_rank"])
    mc, ms = float(np.mean(mean_rank["candidate"])), float(np.mean(mean_rank["symmetric"]))
    print(f"true null, floor removes 2/3: candidate-side selection mean rank {mc:.2f}, rejected at 1 % in {rej['candidate']} of "
          f"{trials} trials; symmetric selection mean rank {ms:.2f}, rejected in {rej['symmetric']}")
    assert mc > 0.62 and rej["candidate"] > 0.10 * trials           # biased: a true null fails >> 1 % of the time
    assert abs(ms - 0.5) < 0.03 and rej["symmetric"] <= 0.03 * trials


def test_wellconditioned_case_oracle_equals_reference():
    """The well-conditioned full-size free-pose case (tests/golden/wc_fullsize_*.npz; workloads.wc_opt_cfg): after 200
    iterations with a FREE pose the oracle and the ACTUAL reference agree to fp32 rounding in state (not merely in
    distribution), the reference's perturbed run stays equally close, and so do the oracle's four perturbed runs on all
    24 instances -- the property that lets tests/test_gpu_fullsize.py demand 1e-4 outright there."""
    ref = np.load(os.path.join(GOLDEN_DIR, "wc_fullsize_reference.npz"))
    orc = np.load(os.path.join(GOLDEN_DIR, "wc_fullsize_oracle.npz"))
    inp = np.load(os.path.join(GOLDEN_DIR, "wc_fullsize_inputs.npz"))
    pos = ref["inst_ids"]
    assert inp["latent0"].shape[0] >= 16 and len(pos) >= 4 and np.all(ref["free_iter_count"] == 200) and np.all(orc["free_iter_count"] == 200)
    zr, Tr = ref["free_latent"], ref["free_T_ow"]
    zo, To = orc["free_latent"], orc["free_T_ow"]
    zscale = np.abs(zo[0]).max(axis=1)                                                    # ~2e-3: the latent does move
    assert np.all(np.abs(zo[0][pos] - zr[0]).max(axis=1) <= 5e-4 * zscale[pos])          # oracle vs reference (measured <= 1.2e-4)
    assert np.all(np.abs(To[0][pos] - Tr[0]).max(axis=(1, 2)) <= 1e-5)                   # (measured <= 1.3e-6)
    assert np.all(np.abs(zr[1:] - zr[0]).max(axis=(0, 2)) <= 2e-3 * zscale[pos])         # reference vs its perturbed run
    assert np.all(np.abs(Tr[1:] - Tr[0]).max(axis=(0, 2, 3)) <= 2e-5)
    assert np.all(np.abs(zo[1:] - zo[0]).max(axis=(0, 2)) <= 2e-3 * zscale)              # oracle vs its 4 perturbed runs, all 24
    assert np.all(np.abs(To[1:] - To[0]).max(axis=(0, 2, 3)) <= 2e-5)
    moved = np.abs(zo[0] - inp["latent0"]).max(axis=1)
    assert np.all(moved > 100 * np.abs(zo[1:] - zo[0]).max(axis=(0, 2)))                 # the optimisation did move the latent
    assert np.median(np.linalg.norm(zo[0], axis=1)) > 3e-3                               # ... comparably to C2 (1.8e-2 free pose)


def test_shape_only_loop_oracle_equals_reference_at_state_level():
    """`shape_opt_deepsdf` (optimizer.py:306-429; bench.py's `c2_sdf`) at L = 256 after 200 forced iterations is WELL
    conditioned: the oracle reproduces the ACTUAL reference's latent to fp32 rounding (measured <= 1e-6 of its size), the
    response of both to a one-ulp scaling of the points is the same smooth 1e-5, and a perturbation of an input this loop
    does not read (foreground depths) reproduces the nominal run bit for bit.  tests/golden/make_sdf_records.py."""
    r = np.load(os.path.join(GOLDEN_DIR, "c2_sdf_fullsize_records.npz"))
    zr, zo = r["ref_latent"], r["orc_latent"]
    n = zr.shape[1]
    sc = np.abs(zo[0]).max(axis=1)
    assert np.all(np.abs(zo[0][:n] - zr[0]).max(axis=1) <= 5e-6 * sc[:n])
    noise_r = np.abs(zr[1:] - zr[0]).max(axis=(0, 2)) / sc[:n]
    noise_o = np.abs(zo[1:3] - zo[0]).max(axis=(0, 2)) / sc
    assert np.all(noise_r <= 1e-4) and np.all(np.abs(noise_o[:n] / noise_r - 1.0) < 0.1)     # the same smooth response
    assert list(r["orc_perts"][3:]) == ["pose0_up", "depth_up"] and np.array_equal(zo[4], zo[0])
    assert np.all(sc > 5e-3)                                                                  # the latent moved
