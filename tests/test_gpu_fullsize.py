"""GPU tests at BASELINE.json's full sizes (64 instances, 256-dim latent, 8x512 decoder, 2048 decoder points per
iteration): size-independent properties (determinism, permutation equivariance, frozen-after-exit) and metric-level
parity (Chamfer-to-ground-truth / pose error) of ALL 64 instances after the full 200 iterations against committed
CPU-oracle records."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L, B = 256, 64
_S = {}


@pytest.fixture(params=["f32", "f16x3", "f16x3f_f16b", "f16"], autouse=True, scope="module")
def precision(request):
    import os
    os.environ["HM_PRECISION"] = request.param
    _S.clear()
    yield request.param
    os.environ.pop("HM_PRECISION", None)
    _S.clear()


def setup():
    if _S:
        return _S
    from hortimapping_amd import synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    params = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(params)
    dicts = W.make_c2_instances(params, dec, list(range(B)), kind="joint")
    _S.update(params=params, dec=dec, dicts=dicts)
    return _S


def run(insts, cfg, shape_only=False):
    from hortimapping_amd import optimizer as HO
    return HO.optimize_batch(setup()["dec"], cfg, insts, shape_only=shape_only)


def test_fullsize_deterministic_and_permutation_equivariant():
    from hortimapping_amd import workloads as W
    s = setup()
    cfg = W.c2_opt_cfg(max_iter=6)
    insts = [W.to_instance(d) for d in s["dicts"]]
    r1 = run(insts, cfg)
    r2 = run(insts, cfg)
    perm = np.random.RandomState(0).permutation(B)
    r3 = run([insts[i] for i in perm], cfg)
    for b in range(B):
        assert r1[b].iter_count == 6 and r1[b].status == 8
        assert torch.isfinite(r1[b].latent).all() and torch.isfinite(r1[b].T_ow).all()
        assert torch.equal(r1[b].latent, r2[b].latent) and torch.equal(r1[b].T_ow, r2[b].T_ow)      # run-to-run bitwise
    for k, i in enumerate(perm):
        assert torch.equal(r3[k].latent, r1[i].latent) and torch.equal(r3[k].T_ow, r1[i].T_ow)      # order only relabels


def test_fullsize_frozen_after_exit(precision):
    """An instance that converges is frozen bit-exactly: running more iterations does not change it."""
    from hortimapping_amd import workloads as W
    if precision == "f16":
        pytest.skip("fp16-class arithmetic: the 3e-4 gradient threshold of this test is below its noise")
    s = setup()
    insts = [W.to_instance(d) for d in s["dicts"][:8]]
    cfg_a = W.c2_opt_cfg(max_iter=12)
    cfg_a["converge"]["epsilon_g"] = 3e-4
    cfg_b = W.c2_opt_cfg(max_iter=20)
    cfg_b["converge"]["epsilon_g"] = 3e-4
    ra, rb = run(insts, cfg_a), run(insts, cfg_b)
    n_conv = 0
    for a, b in zip(ra, rb):
        if a.status & 1:
            n_conv += 1
            assert b.status == a.status and b.iter_count == a.iter_count
            assert torch.equal(a.latent, b.latent) and torch.equal(a.T_ow, b.T_ow)
    assert n_conv > 0


# ----------------------------------------------------------------------------------------------------------------
# Full-batch metric parity at BASELINE.json's size: ALL 64 instances x 200 LM iterations, both pose modes, every
# decoder arithmetic, against CPU-oracle records committed as fixtures (tests/golden/make_fullsize_records.py: the
# oracle on the nominal inputs and on sixteen 1e-7-relative input perturbations, 2,176 runs, ~2.5 h on 8 cores).
# ----------------------------------------------------------------------------------------------------------------
K_NOISE = 3.0          # short-horizon tests: a GPU result may sit K_NOISE x further from the oracle than its perturbed runs
REL_FLOOR = 1e-4       # BASELINE.json north_star: "Chamfer distance / pose error within 1e-4 relative"
# The 200-iteration map is chaotic where the problem is ill conditioned (free pose at the C2 size: the reference itself
# moves by percents under a one-ulp input change; tests/test_fullsize_reference_cpu.py shows the ACTUAL reference and the
# oracle share that noise).  Round 2 bounded every instance by 3 x the largest of its 16 perturbed oracle runs and needed a
# one-outlier allowance -- a weak test with heavy tails.  The gate is now DISTRIBUTIONAL (tests/parity_stats.py): an
# instance inside 1e-4 passes outright; for the others the rank of the GPU deviation among the 16 perturbed-run deviations
# must be uniform over the instances (one-sided Kolmogorov-Smirnov test, alpha = 1e-3: "the GPU arithmetic is no worse than
# a one-ulp input change"), per metric and pose mode, for exact f32 AND f16x3, with no outlier allowance.  K_GROSS is only
# a tripwire for a wrong answer on a single instance (10 x the instance's own band).
# Round 4 (VERDICT / ADVICE of round 3): the rank gate keeps only the instances outside the 1e-4 floor -- 5 to 10 of 64
# in pose_known mode -- and could not tell the mixed mode (Jacobians ~1e-3) from the fp32-class arithmetics there.  It is
# now one of three layers per metric:
#   (1) `parity_stats.exchange_test` over ALL instances: mean log(deviation / largest perturbed deviation) against its
#       Monte-Carlo null "the GPU is one more perturbed run"; the fp32-class arithmetics must pass, and as a POSITIVE
#       CONTROL the mixed and plain-fp16 modes must be REJECTED by the very same statistic in pose_known mode;
#   (2) the KS rank gate, asserted where it has power (its smallest attainable p is below alpha) and reported otherwise;
#   (3) a per-instance cap: no instance beyond K_CAP x its own perturbation band in more than N_CAP_OUT metric entries,
#       none at all beyond K_GROSS x.
# Round 5: the KS layer ranked the instances whose CANDIDATE deviation exceeded the floor -- a
# selection that tilts the surviving ranks towards 1 under a true null (mean rank 0.66 and 29 % rejections at 1 % on
# exchangeable synthetic data, tests/test_fullsize_reference_cpu.py) and produced the "high-side rank bias" of the scale
# error (p = 0.004) for the HIP results and for every one-operation variant of the oracle alike
# (profiles/r05_rank_bias_table.txt).  `parity_stats.gate` now selects symmetrically (largest of all 17 deviations above
# the floor): the HIP ranks come out at 0.49-0.55 with p >= 0.098 in both fp32-class arithmetics and both pose modes, and
# the gates are asserted at alpha = 1e-2.  (The statistics are deterministic -- fixed records, bitwise reproducible HIP
# results -- so alpha is a statement about the margin, not a flake rate: the smallest gated p of this build is 0.045, the
# reference-record Chamfer gate in pose_known mode; the rejected modes sit below 1e-4.)
ALPHA = 1e-2
K_GROSS = 10.0
K_CAP = 4.0
N_CAP_OUT = 1
_FS = {}


RECORDS = {"analytic": "c2_fullsize", "trained": "trained_c2"}      # record set -> fixture prefix under tests/golden/


def trained_params():
    """The decoder learnt by scripts/train_synthetic_deepsdf.py (dense layers, weight-norm g / v, 256 latent codes)."""
    import os
    from golden_util import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "trained_decoder_L256.npz")) as f:
        return {k: (int(f[k]) if k in ("latent_dim", "hidden") else f[k]) for k in f.files}


def fullsize_fixture(which="analytic"):
    """Inputs, oracle records and the per-instance metrics of the oracle runs (computed once per session with ONE
    sampler for every party: the exact-fp32 GPU decoder along 2000 Fibonacci directions, metrics.py)."""
    if which in _FS:
        return _FS[which]
    import os
    from hortimapping_amd import metrics as MX, ops, synthetic as S
    from hortimapping_amd.decoder import DecoderWeights
    from golden_util import GOLDEN_DIR
    inp = np.load(os.path.join(GOLDEN_DIR, RECORDS[which] + "_inputs.npz"))
    rec = np.load(os.path.join(GOLDEN_DIR, RECORDS[which] + "_oracle.npz"))
    params = trained_params() if which == "trained" else S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
    fs = {}
    sampler = DecoderWeights.from_params(params)
    sampler.set_precision("f32")

    def metrics(latents, T_ows):
        """per instance: (Chamfer-to-GT [m], translation error [m], rotation error [deg], scale ratio)"""
        return MX.completion_metrics(sampler, latents, T_ows, fs["gt"], inp["T_wo_true"])
    n = inp["latent0"].shape[0]
    fs["gt"] = MX.ground_truth_points_world(sampler, inp["z_true"], inp["T_wo_true"])
    fs.update(inp=inp, rec=rec, metrics=metrics, n=n, params=params, oracle={})
    for mode in ("known", "free"):
        fs["oracle"][mode] = np.stack([metrics(rec[f"{mode}_latent"][p], rec[f"{mode}_T_ow"][p])
                                        for p in range(rec[f"{mode}_latent"].shape[0])])     # (perts, n, 4)
    _FS[which] = fs
    return fs


def fullsize_instances(pose_known, which="analytic"):
    from hortimapping_amd import optimizer as HO
    inp = fullsize_fixture(which)["inp"]
    t = torch.from_numpy
    return [HO.Instance(t(inp["latent0"][i].copy()), t(inp["T_ow0"][i].copy()), t(inp["points_w"][i]),
                        {"T_wc": [t(inp["T_wc"][i])], "rays_fg": [t(inp["rays_fg"][i])], "rays_bg": [t(inp["rays_bg"][i])],
                         "depth_fg": [t(inp["depth_fg"][i])], "depth_bg": [t(inp["depth_bg"][i])]},
                        float(inp["cube_radius"][i]), pose_known) for i in range(inp["latent0"].shape[0])]


@pytest.mark.parametrize("records", ["analytic", "trained"])
@pytest.mark.parametrize("mode", ["known", "free"])
def test_full_batch_metric_parity(mode, precision, records):
    """`records` = "analytic": the 64 c2_joint instances of the bench (analytic decoder, 16 perturbed oracle runs each);
    "trained": 16 instances of the same workload on the TRAINED decoder (dense layers, 8 perturbed runs each).
    After 200 iterations, per metric m (Chamfer-to-GT as metrics_3d/chamfer_distance.py:16-26 defines it, translation /
    rotation / scale error):  an instance with |m_gpu - m_cpu| <= 1e-4 * scale(m_cpu) passes outright; over the others the
    ranks of |m_gpu - m_cpu| among the instance's perturbed-oracle deviations must pass the one-sided KS test against
    the uniform law (parity_stats.gate).  Gated for the two fp32-class arithmetics (f32, f16x3); the mixed and fp16 modes
    are reported.  The per-instance table goes to gpurun_out/r03_parity_<records>_<mode>_<precision>.txt (-> profiles/)."""
    import os
    import parity_stats as PS
    from hortimapping_amd import optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    fs = fullsize_fixture(records)
    n = fs["n"]
    n_iter = int(fs["rec"]["n_iter"])
    dec = DecoderWeights.from_params(fs["params"])
    dec.set_precision(precision)
    res = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=n_iter), fullsize_instances(mode == "known", records))
    assert all(r.iter_count == n_iter and r.status == 8 for r in res)
    assert np.array_equal(fs["rec"][f"{mode}_iter_count"], np.full_like(fs["rec"][f"{mode}_iter_count"], n_iter))
    m_gpu = fs["metrics"](torch.stack([r.latent for r in res]).numpy(), [r.T_ow.numpy() for r in res])
    _FS.setdefault("gpu", {})[(records, mode, precision)] = m_gpu
    m_all = fs["oracle"][mode]
    m_cpu, m_pert = m_all[0], m_all[1:]
    pert_dev = np.abs(m_pert - m_cpu[None])                                # (K, n, 4)
    noise = pert_dev.max(axis=0)                                           # (n, 4)
    scale = np.stack([m_cpu[:, 0], np.maximum(m_cpu[:, 1], 1e-3), np.maximum(m_cpu[:, 2], 0.1),
                      np.ones(n)], axis=1)                                 # floors: 1 mm, 0.1 deg, unit scale ratio
    floor = REL_FLOOR * scale
    dev = np.abs(m_gpu - m_cpu)
    names = ("chamfer", "t_err", "r_err", "scale")
    gates = {names[k]: PS.gate(dev[:, k], pert_dev[:, :, k], floor[:, k], ALPHA) for k in range(4)}
    exch = {names[k]: PS.exchange_test(dev[:, k], pert_dev[:, :, k]) for k in range(4)}
    gross = [(i, names[k]) for i in range(n) for k in range(4) if dev[i, k] > max(floor[i, k], K_GROSS * noise[i, k])]
    capped = [(i, names[k], round(float(dev[i, k] / max(noise[i, k], 1e-300)), 1)) for i in range(n) for k in range(4)
              if dev[i, k] > max(floor[i, k], K_CAP * noise[i, k])]
    lines = [f"# c2_joint full batch ({records} decoder), {n} instances x {n_iter} LM iterations, pose_{mode}, GPU {precision} "
             "vs CPU oracle",
             f"# gate per metric: inside {REL_FLOOR:g} * scale -> outright; else rank of the deviation among the {m_pert.shape[0]} "
             "perturbed oracle runs of the instance (points x(1+-1e-7), T_ow0 x(1+1e-7), depth_fg x(1+1e-7), independent 1e-7 "
             f"jitters of every point coordinate), one-sided KS over the ranked instances at alpha = {ALPHA:g}",
             "# id  CD_cpu[mm]  CD_gpu[mm]  relCD_gpu  relCD_noise  dT[mm] noise_T[mm]  dR[deg] noise_R[deg]  dS noise_S  rank_CD"]
    ucd = dict(zip(gates["chamfer"]["idx"].tolist(), gates["chamfer"]["u"].tolist()))
    for i in range(n):
        lines.append(f"{i:3d} {1e3 * m_cpu[i, 0]:10.5f} {1e3 * m_gpu[i, 0]:10.5f} {dev[i, 0] / m_cpu[i, 0]:9.2e} "
                     f"{noise[i, 0] / m_cpu[i, 0]:9.2e} {1e3 * dev[i, 1]:9.2e} {1e3 * noise[i, 1]:9.2e} {dev[i, 2]:9.2e} "
                     f"{noise[i, 2]:9.2e} {dev[i, 3]:9.2e} {noise[i, 3]:9.2e}  {('%.2f' % ucd[i]) if i in ucd else 'outright'}")
    relcd, relnoise = dev[:, 0] / m_cpu[:, 0], noise[:, 0] / m_cpu[:, 0]
    lines.append(f"# relative Chamfer-to-GT difference vs the oracle: median {np.median(relcd):.2e} p90 "
                 f"{np.percentile(relcd, 90):.2e} max {relcd.max():.2e};  oracle perturbation noise: median "
                 f"{np.median(relnoise):.2e} p90 {np.percentile(relnoise, 90):.2e} max {relnoise.max():.2e}")
    for k in names:
        g = gates[k]
        lines.append(f"# {k:8s}: within 1e-4 outright {g['outright']:2d} of {n}; ranked {g['ranked']:2d}: mean rank {g['mean_rank']:.2f} "
                     f"(0.5 = like a perturbed run), at the top rank {g['top_rank']}, KS+ {g['ks']:.3f}, p = {g['p']:.3f} "
                     f"(smallest attainable {g['min_p']:.1e}: {'has power' if g['has_power'] else 'NO POWER, reported only'}) "
                     f"-> {'ok' if g['ok'] else 'FAIL'}")
        e = exch[k]
        lines.append(f"# {k:8s}: exchangeability over {e['n']} instances: mean log(dev / max perturbed dev) = {e['T']:.2f}, null "
                     f"{e['null_mean']:.2f} +- {e['null_sd']:.2f} (z = {e['z']:+.1f}), one-sided p = {e['p']:.4f}")
    lines.append(f"# beyond {K_CAP:g} x the instance's own band (allowed: {N_CAP_OUT} entry): {capped}")
    lines.append(f"# beyond {K_GROSS:g} x the instance's own band (gross-error tripwire): {gross}")
    os.makedirs("gpurun_out", exist_ok=True)
    tag = "fullsize" if records == "analytic" else records
    with open(os.path.join("gpurun_out", f"r06_parity_{tag}_{mode}_{precision}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n" + "\n".join(lines[-11:]))
    if precision in ("f32", "f16x3"):          # the fp32-class arithmetics are gated; the other modes are reported
        for k in names:
            assert exch[k]["p"] >= ALPHA, f"{precision} pose_{mode} {k}: worse than a perturbed run ({exch[k]})"
            assert gates[k]["ok"] or not gates[k]["has_power"], \
                f"{precision} pose_{mode} {k}: ranks not uniform (KS+ {gates[k]['ks']:.3f}, p {gates[k]['p']:.2e})"
        assert len(capped) <= N_CAP_OUT, f"{precision} pose_{mode}: beyond {K_CAP} x the instance's own band: {capped}"
        assert not gross, f"{precision} pose_{mode}: beyond {K_GROSS} x the instance's own perturbation band: {gross}"
    elif mode == "known" and records == "analytic":
        # POSITIVE CONTROL: 1e-3 Jacobians (mixed mode) and fp16-class arithmetic must be told from fp32-class by the same
        # statistic that passes f32 / f16x3 above.  (With a free pose the 200-iteration map is chaotic at this size and no
        # statistic of the final state can see them: DESIGN.md section 2.)
        assert exch["chamfer"]["p"] < ALPHA and exch["scale"]["p"] < ALPHA, \
            f"the gate cannot tell {precision} from fp32-class: {exch['chamfer']}, {exch['scale']}"


@pytest.mark.parametrize("records", ["analytic", "trained"])
@pytest.mark.parametrize("mode", ["known", "free"])
def test_fullsize_against_reference_records(mode, precision, records):
    """The same 200-iteration gate against records of the ACTUAL reference loop (tests/golden/c2_fullsize_reference.npz:
    `Optimizer.shape_pose_joint_opt` of /root/reference run on 16 of the 64 instances, nominal + four one-ulp input
    perturbations; generator tests/golden/make_reference_records.py).  Per metric: |m_gpu - m_reference| inside 1e-4
    outright, or its rank among the REFERENCE's own four perturbed deviations uniform over the instances (one-sided KS,
    pooled over the four metrics' ranked entries per instance being correlated, so Chamfer is gated alone and the pose
    metrics are gated together); and the oracle's noise band that calibrates test_full_batch_metric_parity agrees with the
    reference's within 2x (geometric mean).  `records` = "trained": the same on the TRAINED decoder (dense, kinked layers;
    tests/golden/trained_c2_reference.npz: 6 instances, nominal + two perturbed reference runs, band ratio within 3x)."""
    import os
    import parity_stats as PS
    from golden_util import GOLDEN_DIR
    from hortimapping_amd import optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    if precision not in ("f32", "f16x3"):
        pytest.skip("fp32-class arithmetics only")
    fs = fullsize_fixture(records)
    ref = np.load(os.path.join(GOLDEN_DIR, RECORDS[records] + "_reference.npz"))
    ids = ref["inst_ids"]
    m_gpu_all = _FS.get("gpu", {}).get((records, mode, precision))
    if m_gpu_all is None:
        dec = DecoderWeights.from_params(fs["params"])
        dec.set_precision(precision)
        res = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=200), fullsize_instances(mode == "known", records))
        m_gpu_all = fs["metrics"](torch.stack([r.latent for r in res]).numpy(), [r.T_ow.numpy() for r in res])
    sub = {"gt": [fs["gt"][i] for i in ids], "T": fs["inp"]["T_wo_true"][ids]}
    from hortimapping_amd import metrics as MX
    sampler = DecoderWeights.from_params(fs["params"])
    sampler.set_precision("f32")
    m_ref = np.stack([MX.completion_metrics(sampler, ref[f"{mode}_latent"][p], ref[f"{mode}_T_ow"][p], sub["gt"], sub["T"])
                      for p in range(ref[f"{mode}_latent"].shape[0])])                       # (5, 16, 4)
    m_gpu = m_gpu_all[ids]
    m_orc = fs["oracle"][mode][:, ids]                                                          # (17, 16, 4)
    scale = np.stack([m_ref[0][:, 0], np.maximum(m_ref[0][:, 1], 1e-3), np.maximum(m_ref[0][:, 2], 0.1), np.ones(len(ids))], axis=1)
    floor = REL_FLOOR * scale
    pert_ref = np.abs(m_ref[1:] - m_ref[0])                                                    # (4, 16, 4)
    K = m_ref.shape[0] - 1
    nI = len(ids)
    pert_orc = np.abs(m_orc[1:K + 1] - m_orc[0])                                               # the same K perturbations
    out = []
    for who, m in (("gpu", m_gpu), ("oracle", m_orc[0])):
        dev = np.abs(m - m_ref[0])
        g_cd = PS.gate(dev[:, 0], pert_ref[:, :, 0], floor[:, 0], ALPHA)
        g_pose = PS.gate(dev[:, 1:].reshape(-1), pert_ref[:, :, 1:].reshape(K, -1), floor[:, 1:].reshape(-1), ALPHA)
        out.append(f"{who:6s} vs REFERENCE, pose_{mode}: rel CD deviation median {np.median(dev[:, 0] / m_ref[0][:, 0]):.2e} "
                   f"(reference's own noise median {np.median(pert_ref.max(axis=0)[:, 0] / m_ref[0][:, 0]):.2e}); Chamfer: outright "
                   f"{g_cd['outright']}/{nI}, ranked mean {g_cd['mean_rank']:.2f} p {g_cd['p']:.3f}; pose metrics: outright "
                   f"{g_pose['outright']}/{3 * nI}, ranked mean {g_pose['mean_rank']:.2f} p {g_pose['p']:.3f}")
        e_cd = PS.exchange_test(dev[:, 0], pert_ref[:, :, 0])
        # band of the cap: the reference's own K perturbed deviations AND the oracle's for the same perturbations (a max of
        # two or four draws alone is too coarse a band for a per-instance cap)
        band = np.maximum(pert_ref.max(axis=0), pert_orc.max(axis=0))
        over = [(int(ids[i]), k) for i in range(nI) for k in range(4) if dev[i, k] > max(floor[i, k], K_CAP * band[i, k])]
        out.append(f"        Chamfer KS gate: smallest attainable p {g_cd['min_p']:.1e} ({'has power' if g_cd['has_power'] else 'NO power at alpha = %g: the exchangeability test and the cap carry this gate' % ALPHA}); "
                   f"exchangeability p = {e_cd['p']:.4f} (mean log-ratio {e_cd['T']:.2f}, null {e_cd['null_mean']:.2f} +- {e_cd['null_sd']:.2f}); "
                   f"beyond {K_CAP:g} x max(reference band, oracle band): {over}")
        assert (g_cd["ok"] or not g_cd["has_power"]) and (g_pose["ok"] or not g_pose["has_power"]), out[-2]
        assert e_cd["p"] >= ALPHA, out[-1]
        assert len(over) <= N_CAP_OUT, out[-1]
    tiny = 1e-9 * m_ref[0][:, 0]                        # Chamfer noise of the instances, whatever its size
    ratio = float(np.exp(np.mean(np.log((pert_orc.max(axis=0)[:, 0] + tiny) / (pert_ref.max(axis=0)[:, 0] + tiny)))))
    out.append(f"oracle Chamfer noise / reference Chamfer noise over the same {K} perturbations (geometric mean, {nI} instances): {ratio:.2f}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"r06_parity_vs_reference_{records}_{mode}_{precision}.txt"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n" + "\n".join(out))
    lim = 2.0 if K >= 4 else 3.0
    assert 1.0 / lim <= ratio <= lim, ratio


def test_trained_decoder_vs_fp64_oracle(precision):
    """SDF values and input gradients of the TRAINED decoder (dense 512 x 512 layers with learnt weight-norm gains, see
    scripts/train_synthetic_deepsdf.py) against the fp64 oracle, at the learnt codes, on queries around the zero level
    set: the two fp32-class arithmetics to fp32 rounding level, the labelled reduced modes to fp16 level; hidden
    activations stay far inside the fp16 range (the guard would poison the outputs with NaN otherwise)."""
    from hortimapping_amd import ops
    from hortimapping_amd.decoder import DecoderWeights
    from oracle import hm_oracle as O
    from hortimapping_amd import synthetic as S
    tol_y, tol_j, kink = {"f32": (2e-6, 2e-5, 1e-6), "f16x3": (2e-6, 2e-5, 1e-6), "f16x3f_f16b": (2e-6, 5e-3, 1e-6),
                          "f16": (3e-4, 2e-2, 1e-3)}[precision]
    p = trained_params()
    od = O.fold_decoder(p).to(torch.float64)
    dec = DecoderWeights.from_params(p)
    dec.set_precision(precision)
    codes = torch.from_numpy(p["codes"])
    gen = torch.Generator().manual_seed(3)
    B, n = 4, 448
    lat = codes[torch.randperm(codes.shape[0], generator=gen)[:B]].contiguous()
    d = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=gen), dim=-1)
    pts = d * (0.035 + 0.01 * torch.randn(B, n, 1, generator=gen))
    pts4 = torch.zeros(B, n, 4)
    pts4[..., :3] = pts
    y, J = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), torch.full((B,), n, dtype=torch.int32).cuda(), mode=1, pose_dim=0)
    y, J = y.cpu().double(), J.cpu().double()
    Ws, bs = S.fold_weight_norm(p)
    n_kept = 0
    for b in range(B):
        yo, go = O.decoder_jacobian(od, lat[b], pts[b])
        assert torch.isfinite(y[b]).all() and torch.isfinite(J[b]).all()
        assert float(yo.abs().max()) < 0.05 and float((yo < 0).float().mean()) > 0.05      # queries straddle the surface
        assert float((y[b] - yo).abs().max()) < tol_y
        # A trained network has hidden units sitting on their ReLU kink for some query (pre-activations are O(0.1) here
        # and 4096 x 448 of them are evaluated): there the gradient jumps by a finite amount between ANY two arithmetics
        # that round the pre-activation to different signs.  Compare gradients where the fp64 pre-activations keep a
        # distance from zero that fp32 rounding cannot bridge, and demand that this is nearly everywhere.
        u = np.concatenate([np.broadcast_to(lat[b].double().numpy(), (n, L)), pts[b].double().numpy()], axis=1)
        h, margin = u, np.full(n, np.inf)
        for l in range(8):
            if l == 4:
                h = np.concatenate([h, u], axis=1)
            pre = h @ Ws[l].astype(np.float64).T + bs[l].astype(np.float64)
            margin = np.minimum(margin, np.abs(pre).min(axis=1))
            h = np.maximum(pre, 0)
        if precision == "f16":       # fp16 rounding of the activations (~1e-4 absolute) crosses a kink for EVERY query:
            assert float((J[b, :, :L + 3] - torch.cat([go[:, :L], go[:, L:]], 1)).norm() / go.norm()) < 0.05   # norm-wise only
            n_kept += n
            continue
        keep = torch.from_numpy(margin > kink)
        n_kept += int(keep.sum())
        assert float((J[b, keep, :L] - go[keep, :L]).abs().max() / go[:, :L].abs().max()) < tol_j
        assert float((J[b, keep, L:L + 3] - go[keep, L:]).abs().max() / go[:, L:].abs().max()) < tol_j
    assert n_kept >= 0.9 * B * n, n_kept


@pytest.mark.parametrize("mode", ["known", "free"])
def test_trained_short_horizon_parity(mode, precision):
    """The trained-decoder instances after FIVE iterations, before two hundred iterations of a kinked (ReLU) objective
    have amplified every rounding difference: latent and pose of each instance against the oracle record, within
    max(floor, K_NOISE x the deviation of that instance's eight perturbed oracle runs) -- in pose_known mode the
    median bound is ~7e-5 on latent entries of size 0.02.  (Free pose is ill-conditioned from the first solve with
    this decoder: the oracle's own T_ow moves by up to 0.5 under a 1e-7 input perturbation, so that mode only catches
    gross errors.)"""
    import os
    from golden_util import GOLDEN_DIR
    from hortimapping_amd import optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    if precision not in ("f32", "f16x3"):
        pytest.skip("fp32-class arithmetics only")
    rec = np.load(os.path.join(GOLDEN_DIR, "trained_c2_it5_oracle.npz"))
    n_iter = int(rec["n_iter"])
    dec = DecoderWeights.from_params(trained_params())
    dec.set_precision(precision)
    res = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=n_iter), fullsize_instances(mode == "known", "trained"))
    lat = np.stack([r.latent.numpy() for r in res])
    T = np.stack([r.T_ow.numpy() for r in res])
    lat_o, T_o = rec[f"{mode}_latent"], rec[f"{mode}_T_ow"]
    assert all(r.iter_count == n_iter for r in res) and np.all(rec[f"{mode}_iter_count"] == n_iter)
    noise_l = np.abs(lat_o[1:] - lat_o[0]).max(axis=(0, 2))
    noise_T = np.abs(T_o[1:] - T_o[0]).max(axis=(0, 2, 3))
    dl = np.abs(lat - lat_o[0]).max(axis=1)
    dT = np.abs(T - T_o[0]).max(axis=(1, 2))
    bad = [i for i in range(len(res)) if dl[i] > max(1e-5, K_NOISE * noise_l[i]) or dT[i] > max(1e-5, K_NOISE * noise_T[i])]
    print(f"\n{precision} pose_{mode}, {n_iter} iterations: max |d latent| median {np.median(dl):.2e} (oracle noise "
          f"{np.median(noise_l):.2e}), max |d T_ow| median {np.median(dT):.2e} (noise {np.median(noise_T):.2e}); outside: {bad}")
    assert len(bad) <= 1, bad      # short horizon, max-of-8 band: one instance may sit just outside it


# ----------------------------------------------------------------------------------------------------------------
# The well-conditioned full-size case: 1e-4 OUTRIGHT with a free pose (no noise clause, no rank statistics).
# ----------------------------------------------------------------------------------------------------------------
def test_wellconditioned_free_pose_parity(precision):
    """Same sizes as the bench (L = 256, 8 x 512 decoder, 200 forced LM iterations) with a FREE Sim(3) pose, on a case
    where the reference algorithm itself is stable: `workloads.wc_opt_cfg` / `wc_decoder_params` (an elongated fruit, 4 frames
    x 128 rays x 16 samples, render terms weighted 10 x lower, lm_lambda_0 = 1.0 as in lab_berry.yaml) on the instances `tests/golden/make_wc_records.py` kept
    (the candidates whose OWN response to 16 one-ulp input perturbations uses the smallest fraction of the tolerance; the
    stored CPU-oracle records repeat that measurement with four perturbations and it is asserted below).  Gate, for
    EVERY instance and the two fp32-class arithmetics:
        |m_gpu - m_oracle| <= 1e-4 * scale(m_oracle)       m = Chamfer-to-GT, translation / rotation / scale error
    -- BASELINE.json's "Chamfer distance / pose error within 1e-4 relative", outright."""
    import os
    from golden_util import GOLDEN_DIR
    from hortimapping_amd import metrics as MX, optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    f_in, f_rec = os.path.join(GOLDEN_DIR, "wc_fullsize_inputs.npz"), os.path.join(GOLDEN_DIR, "wc_fullsize_oracle.npz")
    inp, rec = np.load(f_in), np.load(f_rec)
    n = inp["latent0"].shape[0]
    n_iter = int(rec["n_iter"])
    assert n >= 16 and n_iter == 200 and np.all(rec["free_iter_count"] == 200)
    params = W.wc_decoder_params(L)
    sampler = DecoderWeights.from_params(params)
    sampler.set_precision("f32")
    gt = MX.ground_truth_points_world(sampler, inp["z_true"], inp["T_wo_true"])
    m_orc = np.stack([MX.completion_metrics(sampler, rec["free_latent"][p], rec["free_T_ow"][p], gt, inp["T_wo_true"])
                      for p in range(rec["free_latent"].shape[0])])
    m_cpu = m_orc[0]
    scale = np.stack([m_cpu[:, 0], np.maximum(m_cpu[:, 1], 1e-3), np.maximum(m_cpu[:, 2], 0.1), np.ones(n)], axis=1)
    tol = REL_FLOOR * scale
    noise = np.abs(m_orc[1:] - m_cpu).max(axis=0)
    # the case IS well conditioned: the oracle's own perturbed runs stay inside half of the tolerance on every instance
    assert np.all(noise <= 0.5 * tol), (noise / tol).max(axis=0)
    dec = DecoderWeights.from_params(params)
    dec.set_precision(precision)
    res = HO.optimize_batch(dec, W.wc_opt_cfg(max_iter=n_iter), [W.to_instance(d, pose_known=False) for d in W.fixture_dicts(inp)])
    assert all(r.iter_count == n_iter and r.status == 8 for r in res)
    m_gpu = MX.completion_metrics(sampler, torch.stack([r.latent for r in res]).numpy(), [r.T_ow.numpy() for r in res], gt,
                                  inp["T_wo_true"])
    dev = np.abs(m_gpu - m_cpu)
    lines = [f"# well-conditioned full-size case, {n} instances x {n_iter} LM iterations, FREE Sim(3) pose, GPU {precision} vs CPU oracle",
             "# id(candidate)  CD_cpu[mm]  relCD_gpu  relCD_oracle_noise   dT[mm]  dR[deg]  dS   (tolerance: 1e-4 x CD, 1e-4 x max(t_err, 1 mm), "
             "1e-4 x max(r_err, 0.1 deg), 1e-4)"]
    for i in range(n):
        lines.append(f"{int(inp['inst_ids'][i]):3d} {1e3 * m_cpu[i, 0]:9.5f} {dev[i, 0] / m_cpu[i, 0]:9.2e} {noise[i, 0] / m_cpu[i, 0]:9.2e} "
                     f"{1e3 * dev[i, 1]:9.2e} {dev[i, 2]:9.2e} {dev[i, 3]:9.2e}  {'ok' if np.all(dev[i] <= tol[i]) else 'FAIL'}")
    lines.append(f"# largest fraction of the 1e-4 tolerance used: GPU {np.max(dev / tol):.2f} (per metric {np.round((dev / tol).max(axis=0), 2).tolist()}), "
                 f"oracle's own perturbed runs {np.max(noise / tol):.2f}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"r06_parity_wellconditioned_free_{precision}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n" + lines[-1])
    # ... and against the ACTUAL reference loop on the first instances (tests/golden/make_reference_records.py --case wc:
    # `Optimizer.shape_pose_joint_opt` of /root/reference, nominal + one perturbed run each)
    ref = np.load(os.path.join(GOLDEN_DIR, "wc_fullsize_reference.npz"))
    pos = ref["inst_ids"]                                   # positions in the inputs fixture
    m_ref = np.stack([MX.completion_metrics(sampler, ref["free_latent"][p], ref["free_T_ow"][p], [gt[i] for i in pos],
                                            inp["T_wo_true"][pos]) for p in range(ref["free_latent"].shape[0])])
    assert np.all(ref["free_iter_count"] == n_iter)
    d_or, d_gr, d_rr = np.abs(m_cpu[pos] - m_ref[0]), np.abs(m_gpu[pos] - m_ref[0]), np.abs(m_ref[1:] - m_ref[0]).max(axis=0)
    line = (f"# vs the ACTUAL reference on {len(pos)} instances, fraction of the 1e-4 tolerance used: oracle {np.max(d_or / tol[pos]):.2f}, "
            f"GPU {precision} {np.max(d_gr / tol[pos]):.2f}, the reference's own perturbed run {np.max(d_rr / tol[pos]):.2f}")
    print(line)
    with open(os.path.join("gpurun_out", f"r06_parity_wellconditioned_free_{precision}.txt"), "a") as f:
        f.write(line + "\n")
    assert np.all(d_or <= tol[pos]) and np.all(d_rr <= tol[pos])     # oracle == reference, and the reference is stable here
    if precision in ("f32", "f16x3"):
        assert np.all(dev <= tol), [(int(inp["inst_ids"][i]), (dev[i] / tol[i]).round(2).tolist()) for i in range(n) if np.any(dev[i] > tol[i])]
        assert np.all(d_gr <= tol[pos])
    else:
        # POSITIVE CONTROL: the outright 1e-4 gate must reject the arithmetics that are not fp32-class (measured: the mixed
        # mode uses 1.4 x the tolerance, plain fp16 320 x)
        assert not np.all(dev <= tol), f"{precision} passes the fp32-class gate: the gate has no power"


def test_wellconditioned_pass_fraction_over_all_candidates(precision):
    """Sizes the outright-1e-4 claim honestly (round-3 review): `test_wellconditioned_free_pose_parity` runs on the 23 of
    128 candidate instances whose own response to one-ulp input perturbations is smallest (a selection on CONDITIONING, made
    with the HIP f32 path: tests/golden/wc_selection.json).  Here ALL 128 candidates are optimised and compared with the
    CPU oracle's nominal run (tests/golden/wc_all_oracle.npz, `make_wc_records.py all_nominal`); the fraction that meets
    1e-4 outright is REPORTED next to the fraction whose own one-ulp response stays inside the tolerance -- on an
    ill-conditioned candidate no two arithmetics (not even two runs of the reference on inputs one ulp apart) agree to
    1e-4, so the first fraction cannot exceed the second by much.  Asserted: the selected instances are among the passing
    ones, and an fp32-class arithmetic passes wherever the instance itself is stable to a third of the tolerance."""
    import json
    import os
    from golden_util import GOLDEN_DIR
    from hortimapping_amd import metrics as MX, optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    f_all = os.path.join(GOLDEN_DIR, "wc_all_oracle.npz")
    if not os.path.exists(f_all):
        pytest.skip("tests/golden/wc_all_oracle.npz not generated")
    if precision not in ("f32", "f16x3"):
        pytest.skip("fp32-class arithmetics only")
    cand, rec = np.load(os.path.join(GOLDEN_DIR, "wc_candidates.npz")), np.load(f_all)
    sel = json.load(open(os.path.join(GOLDEN_DIR, "wc_selection.json")))
    kept_ids = set(np.load(os.path.join(GOLDEN_DIR, "wc_fullsize_inputs.npz"))["inst_ids"].tolist())
    n = cand["latent0"].shape[0]
    assert rec["free_latent"].shape[0] == n and np.array_equal(rec["inst_ids"], cand["inst_ids"])
    params = W.wc_decoder_params(L)
    sampler = DecoderWeights.from_params(params).set_precision("f32")
    gt = MX.ground_truth_points_world(sampler, cand["z_true"], cand["T_wo_true"])
    m_cpu = MX.completion_metrics(sampler, rec["free_latent"], rec["free_T_ow"], gt, cand["T_wo_true"])
    dec = DecoderWeights.from_params(params).set_precision(precision)
    res = HO.optimize_batch(dec, W.wc_opt_cfg(max_iter=200), [W.to_instance(d, pose_known=False) for d in W.fixture_dicts(cand)])
    assert all(r.iter_count == 200 and r.status == 8 for r in res)
    m_gpu = MX.completion_metrics(sampler, torch.stack([r.latent for r in res]).numpy(), [r.T_ow.numpy() for r in res], gt, cand["T_wo_true"])
    scale = np.stack([m_cpu[:, 0], np.maximum(m_cpu[:, 1], 1e-3), np.maximum(m_cpu[:, 2], 0.1), np.ones(n)], axis=1)
    used = (np.abs(m_gpu - m_cpu) / (REL_FLOOR * scale)).max(axis=1)              # fraction of the tolerance used, per candidate
    own = np.array(sel["score"])                                                   # the candidate's own one-ulp response / tolerance
    ok = used <= 1.0
    kept = np.array([int(i) in kept_ids for i in cand["inst_ids"]])
    lines = [f"# well-conditioned CONFIGURATION, all {n} candidate instances x 200 LM iterations, free Sim(3) pose, GPU {precision} vs the CPU oracle's nominal run",
             f"# meet 1e-4 outright (all four metrics): {int(ok.sum())} of {n} = {ok.mean():.2f};  of the {int(kept.sum())} selected: {int(ok[kept].sum())}",
             f"# candidates whose OWN response to 16 one-ulp input perturbations (HIP f32, wc_selection.json) stays inside the tolerance: "
             f"{int((own <= 1.0).sum())} of {n} = {(own <= 1.0).mean():.2f}; inside a third of it: {int((own <= 1 / 3).sum())}",
             f"# GPU-vs-oracle tolerance use: median {np.median(used):.2f}, p90 {np.percentile(used, 90):.2f}, max {used.max():.1f};  own response: median {np.median(own):.2f}, p90 {np.percentile(own, 90):.2f}",
             "# id  own_response/tol  gpu_vs_oracle/tol  selected"]
    lines += [f"{int(cand['inst_ids'][i]):3d} {own[i]:8.2f} {used[i]:8.2f}  {'*' if kept[i] else ''}" for i in range(n)]
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"r06_parity_wellconditioned_all_candidates_{precision}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n" + "\n".join(lines[:4]))
    assert np.all(ok[kept])
    # ... and wherever the instance itself is stable to a third of the tolerance -- except the candidates whose CPU-ORACLE runs
    # spread by more than half the tolerance among themselves (wc_fixture_report.json: candidate 17, a second basin that the
    # oracle, f32 and f16x3 all see: `make_wc_records.py prune`)
    rep = json.load(open(os.path.join(GOLDEN_DIR, "wc_fixture_report.json")))
    oracle_unstable = {i for i, f in zip(rep["inst_ids"], rep["oracle_noise_frac"]) if f > 0.5}
    stable = (own <= 1.0 / 3.0) & ~np.array([int(i) in oracle_unstable for i in cand["inst_ids"]])
    assert np.all(ok[stable]), [(int(cand["inst_ids"][i]), float(own[i]), float(used[i])) for i in np.nonzero(stable & ~ok)[0]]


@pytest.mark.parametrize("npts", [1024, 2048])
def test_shape_only_fullsize_state_parity(precision, npts):
    """The shape-only loop (`shape_opt_deepsdf`, bench.py's `c2_sdf` line) at full size -- L = 256, 200 forced iterations,
    16 instances -- is well conditioned, so parity is asserted at STATE level and outright: the HIP latent against the CPU
    oracle AND against the records of the ACTUAL reference loop (oracle == reference to 1e-6 there), to 5e-5 / 1e-5 of the
    latent's size in the two fp32-class arithmetics; the pose must come back untouched.
    npts = 1024: the C2 fixture's surface points (tests/golden/c2_sdf_fullsize_records.npz); npts = 2048: the instances of
    the `c2_sdf` BENCH workload itself, BASELINE.json's literal "2048 pts/instance" (c2_sdf2048_inputs.npz / _records.npz,
    `make_sdf_records.py 2048`).  POSITIVE CONTROL: the mixed and fp16 modes must FAIL the fp32-class bound (they land at
    1e-4 and 1e-3)."""
    import os
    from golden_util import GOLDEN_DIR
    from hortimapping_amd import optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    fs = fullsize_fixture("analytic")
    rec = np.load(os.path.join(GOLDEN_DIR, "c2_sdf_fullsize_records.npz" if npts == 1024 else f"c2_sdf{npts}_records.npz"))
    zo, zr = rec["orc_latent"][0], rec["ref_latent"][0]
    n = zo.shape[0]
    dec = DecoderWeights.from_params(fs["params"])
    dec.set_precision(precision)
    if npts == 1024:
        insts = fullsize_instances(False, "analytic")[:n]
    else:
        inp = np.load(os.path.join(GOLDEN_DIR, f"c2_sdf{npts}_inputs.npz"))
        t = torch.from_numpy
        assert inp["points_w"].shape[1:] == (npts, 3)
        insts = [HO.Instance(t(inp["latent0"][i].copy()), t(inp["T_ow0"][i].copy()), t(inp["points_w"][i]), None, 0.08, True)
                 for i in range(n)]
    res = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=200), insts, shape_only=True)
    assert all(r.iter_count == 200 and r.status == 8 for r in res)
    z = np.stack([r.latent.numpy() for r in res])
    sc = np.abs(zo).max(axis=1)
    d_o = np.abs(z - zo).max(axis=1) / sc
    d_r = np.abs(z[:zr.shape[0]] - zr).max(axis=1) / sc[:zr.shape[0]]
    print(f"\nshape-only, {npts} points, {precision}: max |z_gpu - z_oracle| / max|z|: median {np.median(d_o):.2e} max {d_o.max():.2e}; vs the reference: max {d_r.max():.2e}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"r06_parity_shape_only_{npts}_{precision}.txt"), "w") as f:
        f.write(f"# shape_opt_deepsdf, 16 instances x 200 iterations, L = 256, {npts} surface points, GPU {precision}: relative latent deviation per instance\n"
                "# id  vs_oracle  vs_reference(first 8)\n" +
                "\n".join(f"{i:2d} {d_o[i]:.2e} {(d_r[i] if i < len(d_r) else float('nan')):.2e}" for i in range(n)) + "\n")
    for inst, r in zip(insts, res):
        assert torch.equal(r.T_ow, inst.T_ow)
    if precision in ("f32", "f16x3"):
        # measured (1024 points): vs the reference <= 2.2e-6 (8 instances), vs the oracle median 2e-6, max 1.9e-5 (one of 16)
        assert d_o.max() < 5e-5 and d_r.max() < 1e-5
    else:
        assert d_o.max() > 5e-5 or d_r.max() > 1e-5, f"{precision} passes the fp32-class bound: the gate has no power"
