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
K_NOISE = 3.0          # a GPU result may sit K_NOISE x further from the oracle than the oracle's own perturbed runs
REL_FLOOR = 1e-4       # BASELINE.json north_star: "Chamfer distance / pose error within 1e-4 relative"
# The 200-iteration map is chaotic: every arithmetic (and every change of summation order, solver, tile order) is one more
# draw from the heavy-tailed distribution the 16 perturbed oracle runs sample.  A 17th independent draw exceeds 3 x the
# maximum of 16 with probability ~1 % per instance (measured: exact f32 1 of 64 in each mode, f16x3 0, with the build
# these records were first used on; a different build moves WHICH instance), so the gate allows ONE instance per mode
# beyond K_NOISE, held to K_OUTLIER instead, and prints it.
N_OUTLIER = 1
K_OUTLIER = 6.0
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
    For EVERY instance after 200 iterations:
        |m_gpu - m_cpu| <= max(1e-4 * scale(m_cpu), K_NOISE * noise_i)        m = Chamfer-to-GT, pose errors
    with noise_i = the largest deviation of the sixteen perturbed oracle runs of instance i from its nominal run (the
    reference algorithm's own response to a 1e-7 relative input change).  The two fp32-class arithmetics (f32, f16x3)
    must pass on all 64 but N_OUTLIER, which is held to K_OUTLIER x noise_i (see the constants); every instance outside
    the K_NOISE bound is listed by id in the table and in the assertion message.  The per-instance table is written to
    gpurun_out/r02_parity_fullsize_<mode>_<precision>.txt (copied to profiles/)."""
    import os
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
    m_all = fs["oracle"][mode]
    m_cpu, m_pert = m_all[0], m_all[1:]
    noise = np.abs(m_pert - m_cpu[None]).max(axis=0)                       # (n, 4)
    scale = np.stack([m_cpu[:, 0], np.maximum(m_cpu[:, 1], 1e-3), np.maximum(m_cpu[:, 2], 0.1),
                      np.ones(n)], axis=1)                                 # floors: 1 mm, 0.1 deg, unit scale ratio
    tol = np.maximum(REL_FLOOR * scale, K_NOISE * noise)
    tol_out = np.maximum(REL_FLOOR * scale, K_OUTLIER * noise)
    dev = np.abs(m_gpu - m_cpu)
    names = ("chamfer", "t_err", "r_err", "scale")
    lines = [f"# c2_joint full batch ({records} decoder), {n} instances x {n_iter} LM iterations, pose_{mode}, GPU {precision} "
             "vs CPU oracle",
             f"# tolerance per instance and metric: max({REL_FLOOR:g} * scale, {K_NOISE:g} * noise_i); noise_i = max deviation "
             f"of {m_pert.shape[0]} perturbed oracle runs (points x(1+-1e-7), T_ow0 x(1+1e-7), depth_fg x(1+1e-7), "
             "independent 1e-7 jitters of every point coordinate)",
             "# id  CD_cpu[mm]  CD_gpu[mm]  relCD_gpu  relCD_noise  dT[mm] noise_T[mm]  dR[deg] noise_R[deg]  dS noise_S  verdict"]
    bad = []
    for i in range(n):
        ok = bool((dev[i] <= tol[i]).all())
        if not ok:
            bad.append((i, [names[k] for k in range(4) if dev[i, k] > tol[i, k]]))
        lines.append(f"{i:3d} {1e3 * m_cpu[i, 0]:10.5f} {1e3 * m_gpu[i, 0]:10.5f} {dev[i, 0] / m_cpu[i, 0]:9.2e} "
                     f"{noise[i, 0] / m_cpu[i, 0]:9.2e} {1e3 * dev[i, 1]:9.2e} {1e3 * noise[i, 1]:9.2e} {dev[i, 2]:9.2e} "
                     f"{noise[i, 2]:9.2e} {dev[i, 3]:9.2e} {noise[i, 3]:9.2e}  {'ok' if ok else 'FAIL'}")
    relcd, relnoise = dev[:, 0] / m_cpu[:, 0], noise[:, 0] / m_cpu[:, 0]
    lines.append(f"# relative Chamfer-to-GT difference vs the oracle: median {np.median(relcd):.2e} p90 "
                 f"{np.percentile(relcd, 90):.2e} max {relcd.max():.2e};  oracle perturbation noise: median "
                 f"{np.median(relnoise):.2e} p90 {np.percentile(relnoise, 90):.2e} max {relnoise.max():.2e}")
    lines.append(f"# instances within 1e-4 relative Chamfer outright: {(relcd <= 1e-4).sum()} of {n}; failing the gate: "
                 f"{[b[0] for b in bad]}")
    os.makedirs("gpurun_out", exist_ok=True)
    tag = "fullsize" if records == "analytic" else records
    with open(os.path.join("gpurun_out", f"r02_parity_{tag}_{mode}_{precision}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[-2:]))
    if precision in ("f32", "f16x3"):          # the fp32-class arithmetics are gated; the other modes are reported
        assert len(bad) <= N_OUTLIER, f"{precision} pose_{mode}: instances outside max(1e-4, {K_NOISE} x noise): {bad}"
        for i, _ in bad:
            assert bool((dev[i] <= tol_out[i]).all()), f"{precision} pose_{mode}: instance {i} beyond {K_OUTLIER} x noise"


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
    assert len(bad) <= N_OUTLIER, bad
