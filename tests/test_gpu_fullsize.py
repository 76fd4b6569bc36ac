"""GPU tests at BASELINE.json's full sizes (64 instances, 256-dim latent, 8x512 decoder, 2048 decoder points per
iteration) through size-independent properties, plus metric-level parity (Chamfer / pose error) against the oracle
on a sample of the batch.  The iteration count is reduced so the whole file runs in about a minute."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L, B = 256, 64
_S = {}
_ORACLE = {}      # CPU oracle results, shared by the two precision parametrisations


@pytest.fixture(params=["f32", "f16x3"], autouse=True, scope="module")
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


def test_fullsize_frozen_after_exit():
    """An instance that converges is frozen bit-exactly: running more iterations does not change it."""
    from hortimapping_amd import workloads as W
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


@pytest.mark.parametrize("pose_known", [True, False])
def test_fullsize_metric_parity_vs_oracle(pose_known):
    """Chamfer-to-ground-truth and pose error of the HIP result vs the CPU oracle's result on the same inputs
    (two instances of the batch, 20 LM iterations).  BASELINE.json asks for 1e-4 relative: pose_known runs are held
    to it.  Free-pose runs are chaotic (hard with_grad / ball / Huber / ReLU switches amplify rounding noise, SURVEY.md
    8d): two fp32 evaluations of the REFERENCE ALGORITHM ITSELF (the oracle vs the oracle with the surface points
    scaled by 1 + 1e-7) differ by 3e-4 ... 1e-2 in Chamfer-to-GT at these sizes (scripts/parity_fullsize.py, DESIGN.md
    section 2), and which instance flips is perturbation dependent.  The free-pose bar is therefore that band (2e-2),
    or 5x the noise measured in this very test if that is larger."""
    from hortimapping_amd import metrics as MX, utils as U, workloads as W
    from oracle import hm_oracle as O
    s = setup()
    n_it = 20
    cfg = W.c2_opt_cfg(max_iter=n_it)
    pick = [3, 41]
    insts = [W.to_instance(s["dicts"][i], pose_known=pose_known) for i in range(B)]
    res = run(insts, cfg)
    od = O.fold_decoder(s["params"])
    dec = s["dec"]

    def pts(latent, T_ow):
        return MX.completed_points_world(lambda p: U.decode_sdf(dec, latent, torch.from_numpy(p)).cpu().numpy(),
                                         T_ow.numpy())
    for i in pick:
        d = s["dicts"][i]
        rd = {k: [torch.from_numpy(a) for a in v] for k, v in d["render"].items()}
        args = (od, cfg, torch.from_numpy(d["latent0"]), torch.from_numpy(d["T_ow0"]), rd)
        if (i, pose_known) not in _ORACLE:
            a = O.shape_pose_joint_opt(*args, torch.from_numpy(d["points_w"]), d["cube_radius"], pose_known=pose_known)
            b2 = O.shape_pose_joint_opt(*args, torch.from_numpy(d["points_w"]) * (1 + 1e-7), d["cube_radius"],
                                        pose_known=pose_known)
            _ORACLE[(i, pose_known)] = (a, b2)
        (z, T, n), (z2, T2, _) = _ORACLE[(i, pose_known)]
        assert res[i].iter_count == n == n_it
        gt = pts(torch.from_numpy(d["z_true"]), torch.from_numpy(np.linalg.inv(d["T_wo_true"]).astype(np.float32)))
        cd_gpu = MX.chamfer_distance(pts(res[i].latent, res[i].T_ow), gt)
        cd_cpu = MX.chamfer_distance(pts(z, T), gt)
        cd_cpu2 = MX.chamfer_distance(pts(z2, T2), gt)
        rel = abs(cd_gpu - cd_cpu) / cd_cpu
        noise = abs(cd_cpu2 - cd_cpu) / cd_cpu
        pe_g, pe_c = MX.pose_error(res[i].T_ow.numpy(), d["T_wo_true"]), MX.pose_error(T.numpy(), d["T_wo_true"])
        pe_c2 = MX.pose_error(T2.numpy(), d["T_wo_true"])
        if pose_known:
            tol_cd, tol_t, tol_s = 1e-4, 1e-4 * max(pe_c[0], 1e-3), 1e-4
        else:
            tol_cd = max(2e-2, 5 * noise)
            tol_t = max(2e-2 * max(pe_c[0], 1e-3), 5 * abs(pe_c2[0] - pe_c[0]))
            tol_s = max(2e-2, 5 * abs(pe_c2[2] - pe_c[2]))
        assert rel < tol_cd, (rel, noise, cd_gpu, cd_cpu)
        assert abs(pe_g[0] - pe_c[0]) < tol_t
        assert abs(pe_g[2] - pe_c[2]) < tol_s
