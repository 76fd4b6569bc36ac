"""Round-5 GPU tests (run with `-m gpu`): behaviour added or un-deaded this round.  No module-level HM_PRECISION fixture
here ON PURPOSE -- the first test needs the process default the drop-in class sees in a user's script."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _overflow_params():
    """A decoder that is the SAME function in fp32 but whose layer-1 activations (~1e5) leave the fp16 range."""
    from hortimapping_amd import synthetic as S
    p = S.make_synthetic_decoder(32, seed=3, r0=0.04, aniso=(1.0, 0.75, 1.3))
    big = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in p.items()}
    big["lin1.weight_g"] = big["lin1.weight_g"] * 4.0e6
    big["lin2.weight_g"] = big["lin2.weight_g"] / 4.0e6
    return p, big


def _module(params, L=32):
    from hortimapping_amd import synthetic as S

    class Net(torch.nn.Module):                       # the reference passes an nn.Module (optimizer.py:17)
        def __init__(self):
            super().__init__()
            for l, (o_, i_) in enumerate(S.layer_shapes(L)):
                lin = torch.nn.Linear(i_, o_)
                setattr(self, f"lin{l}", torch.nn.utils.weight_norm(lin) if l < 8 else lin)
    net = Net()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in params.items()
                         if k not in ("latent_dim", "hidden")}, strict=True)
    return net


def test_optimizer_from_module_defaults_to_f16x3_and_retries_by_itself(monkeypatch):
    """`Optimizer(cfg, nn.Module)` with no HM_PRECISION in the environment (a user's script): the class selects f16x3 and
    its own `retry_f32` turns an out-of-fp16-range decoder into the pure exact-f32 result, bit for bit; the decoder handle
    the class owns keeps its f16x3 setting (the retry runs on the f32 twin)."""
    monkeypatch.delenv("HM_PRECISION", raising=False)
    from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p, big = _overflow_params()
    Ws, bs = S.fold_weight_norm(p)
    insts = [W.to_instance(S.make_instance(Ws, bs, 32, i, n_pts=256, n_frames=1, n_fg=16, n_bg=16), pose_known=True)
             for i in (1, 2)]
    opt = W.c2_opt_cfg(max_iter=3)
    pure = HO.optimize_batch(DecoderWeights.from_params(big).set_precision("f32"), opt, insts)
    assert all(r.status == 8 and r.iter_count == 3 for r in pure)
    o = HO.Optimizer({"device": "cuda", "opt": opt, "vis": {}}, _module(big))
    assert o.decoder.precision == "f16x3"
    for k, inst in enumerate(insts):
        lat = inst.latent.clone()
        z, T, n = o.shape_pose_joint_opt(lat, inst.T_ow, inst.render_data, inst.points_w, inst.cube_radius, None, True)
        assert z is lat and n == 3
        assert torch.equal(z.cpu(), pure[k].latent) and torch.equal(T.cpu(), pure[k].T_ow)
        assert o.decoder.precision == "f16x3"
    res = o.optimize_batch(insts)
    assert all(r.retried_f32 for r in res) and all(torch.equal(a.latent, b.latent) for a, b in zip(res, pure))
    # an in-range decoder through the same class: f16x3 results, nothing retried
    o2 = HO.Optimizer({"device": "cuda", "opt": opt, "vis": {}}, _module(p))
    r2 = o2.optimize_batch(insts)
    ref = HO.optimize_batch(DecoderWeights.from_params(p).set_precision("f16x3"), opt, insts)
    assert all((not a.retried_f32) and torch.equal(a.latent, b.latent) for a, b in zip(r2, ref))


def test_mesh_grid_decode_falls_back_to_f32_per_instance():
    """the entry points switch the shared decoder to f16x3 and the final voxel-grid decode ran in it too; a
    poisoned tile would have put NaN into marching cubes and the .ply.  `MeshExtractor.decode_grids` now re-decodes the
    instances with non-finite values in exact fp32 (on the f32 twin): the grid equals the pure-f32 grid bit for bit."""
    from hortimapping_amd.decoder import DecoderWeights
    from hortimapping_amd.mesher import MeshExtractor
    p, big = _overflow_params()
    lat = 0.05 * torch.randn(3, 32, generator=torch.Generator().manual_seed(1))
    d32 = DecoderWeights.from_params(big).set_precision("f32")
    g32 = MeshExtractor(d32, code_len=32, voxels_dim=24, cube_radius=0.08).decode_grids(lat)
    dh = DecoderWeights.from_params(big).set_precision("f16x3")
    mx = MeshExtractor(dh, code_len=32, voxels_dim=24, cube_radius=0.08)
    gh = mx.decode_grids(lat)
    assert torch.isfinite(gh).all() and torch.equal(gh, g32) and mx.n_f32_redecoded == 3
    assert dh.precision == "f16x3"
    meshes = mx.extract_meshes(lat)
    assert all(m.vertices.shape[0] > 0 and np.isfinite(m.vertices).all() for m in meshes)
    # in range: the f16x3 grid is used as it is
    ok = DecoderWeights.from_params(p).set_precision("f16x3")
    mo = MeshExtractor(ok, code_len=32, voxels_dim=24, cube_radius=0.08)
    assert torch.isfinite(mo.decode_grids(lat)).all() and getattr(mo, "n_f32_redecoded", 0) == 0


_PACING_INSTANCES = []      # generated once (numpy forward, ~30 s): the instances do not depend on the arithmetic under test


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_host_pacing_off_is_a_pure_enqueue_with_the_same_bits(precision):
    """hm_workspace_set_host_pacing(ws, 0): with early exits possible the call must not wait for the device (it is then
    legal under stream capture) and must give the same bits as the paced call."""
    import time
    from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p = S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
    Ws, bs = S.fold_weight_norm(p)
    dec = DecoderWeights.from_params(p).set_precision(precision)
    if not _PACING_INSTANCES:
        fac = W.gpu_sdf_factory(DecoderWeights.from_params(p).set_precision("f32"))   # (the numpy forward took 30 s here)
        _PACING_INSTANCES.extend(S.make_instance(Ws, bs, 32, i, n_pts=512, n_frames=2, n_fg=64, n_bg=64, sdf_fn_factory=fac)
                                 for i in range(20))
    insts = [W.to_instance(d) for d in _PACING_INSTANCES]
    from oracle import hm_oracle as O
    opt = O.default_opt_cfg()                       # wild_pepper.yaml block: every epsilon > 0, max_iter 50
    opt["render"]["n_frame"] = 2
    a = HO.optimize_batch(dec, opt, insts)
    ws = HO.Workspace(dec, 20, 512, 2, 128, 30).set_host_pacing(False)
    torch.cuda.synchronize()
    pb = HO.PackedBatch(insts, 32, 2, "cuda", F_cap=2, R_cap=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    HO.run_packed(ws, HO.opt_cfg_from_dict(opt), pb, 0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert sum(1 < r.iter_count < 50 for r in a) >= 10, [r.iter_count for r in a]     # early exits really happen
    for b, r in enumerate(a):
        assert int(pb.iter_count[b]) == r.iter_count and int(pb.status[b]) == r.status
        assert torch.equal(pb.latent[b].cpu(), r.latent) and torch.equal(pb.T_ow[b].cpu().reshape(4, 4), r.T_ow)
    print(f"unpaced enqueue {t_enq * 1e3:.1f} ms of {t_all * 1e3:.1f} ms total")
    ws.release()


def _linear_cfg_instances(yaml_name, n, shape, pose_known, L=32, r0=0.04):
    import yaml
    from hortimapping_amd import synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt = yaml.safe_load(open(os.path.join(root, "configs", yaml_name)))["opt"]
    p = S.make_synthetic_decoder(L, seed=1, r0=r0, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(p)
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)
    insts = [W.to_instance(S.make_instance(Ws, bs, L, i, sdf_fn_factory=fac, **shape), pose_known=pose_known)
             for i in range(n)]
    return opt, dec, insts


@pytest.mark.parametrize("precision", ["f16x3", "f16x3f_f16b"])
@pytest.mark.parametrize("case", ["challenge", "lab_pepper"])
def test_linear_occupancy_screening_gives_the_same_bits(case, precision):
    """Round 5: under LINEAR occupancy the ball-valid ray samples are screened by one fp16 pass and
    only the ones that may lie in the +-cutoff band (plus margin) go through the f16x3 forward.  The whole optimisation
    must come out BIT-IDENTICAL to the unscreened run (latent, pose, iteration count, status), the verify mode must find
    no screened-far sample whose exact sdf is in the band, and most samples must in fact be screened away."""
    from hortimapping_amd import optimizer as HO
    if case == "challenge":
        opt, dec, insts = _linear_cfg_instances("shape_completion_challenge_pepper.yaml", 24,
                                                dict(n_pts=2000, n_frames=5, n_fg=200, n_bg=100), True)
    else:
        opt, dec, insts = _linear_cfg_instances("lab_pepper.yaml", 24, dict(n_pts=2000, n_frames=5, n_fg=200, n_bg=100), False)
    assert not opt["render"]["log_sdf_occ"]
    dec.set_precision(precision)
    hcfg = HO.opt_cfg_from_dict(opt)
    L = dec.latent_dim
    out = {}
    for mode in (0, 1, 2, 3):
        pb = HO.PackedBatch(insts, L, int(opt["render"]["n_frame"]), "cuda")
        ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray).set_screening(mode)
        if mode >= 2:
            ws.screening_stats(reset=True)
        HO.run_packed(ws, hcfg, pb, 0)
        torch.cuda.synchronize()
        out[mode] = (pb.latent.cpu(), pb.T_ow.cpu(), pb.iter_count.cpu(), pb.status.cpu())
        if mode == 2:
            st = ws.screening_stats()
        if mode == 3:                      # round 6: verify in the first iteration only (what the Python host runs once per handle)
            st3 = ws.screening_stats()
        ws.release()
    for mode in (1, 2, 3):
        for a, b in zip(out[0], out[mode]):
            assert torch.equal(a, b), f"screening mode {mode} changed the result"
    assert st3["violations"] == 0 and 0 < st3["screened"] < st["screened"], (st3, st)     # one iteration's samples, not the call's
    assert torch.isfinite(out[0][0]).all() and int(out[0][2].min()) >= 1
    print(case, precision, st)
    assert st["violations"] == 0
    assert st["screened"] > 0 and st["promoted"] < 0.5 * st["screened"], st


def test_screening_is_off_for_logistic_occupancy_and_exact_f32():
    """The screening must not touch configurations it does not apply to: logistic occupancy (wild_pepper.yaml) and the
    exact-f32 / plain-fp16 arithmetics never promote or screen anything (stats stay zero in verify mode)."""
    from hortimapping_amd import optimizer as HO
    opt, dec, insts = _linear_cfg_instances("wild_pepper.yaml", 4, dict(n_pts=500, n_frames=2, n_fg=100, n_bg=100), False)
    assert opt["render"]["log_sdf_occ"]
    hcfg = HO.opt_cfg_from_dict(opt)
    for prec in ("f16x3", "f32"):
        dec.set_precision(prec)
        pb = HO.PackedBatch(insts, 32, int(opt["render"]["n_frame"]), "cuda")
        ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray).set_screening(2)
        ws.screening_stats(reset=True)
        HO.run_packed(ws, hcfg, pb, 0)
        assert ws.screening_stats() == {"screened": 0, "promoted": 0, "violations": 0, "dead": 0}
        ws.release()
    opt2, dec2, insts2 = _linear_cfg_instances("lab_pepper.yaml", 4, dict(n_pts=500, n_frames=2, n_fg=100, n_bg=100), False)
    dec2.set_precision("f32")
    pb = HO.PackedBatch(insts2, 32, int(opt2["render"]["n_frame"]), "cuda")
    ws = HO.Workspace(dec2, pb.B, pb.points_stride, pb.F, pb.R, HO.opt_cfg_from_dict(opt2).n_sample_on_ray).set_screening(2)
    ws.screening_stats(reset=True)
    HO.run_packed(ws, HO.opt_cfg_from_dict(opt2), pb, 0)
    assert ws.screening_stats()["screened"] == 0
    ws.release()


def test_normal_equations_on_the_fp16_matrix_cores_agree_with_the_fp32_kernel():
    """K4h (round 5): in the f16x3 arithmetics the normal equations run on the fp16 matrix cores with split operands.  Same
    rows in, so H and b must agree with the fp32-input kernel to fp32 rounding (products exact, 2^-22 dropped term) -- checked on
    the damped system of a full-size C2-joint iteration (L = 256, 1024 + V + V rows, Huber weights on) and on a wild_pepper-sized
    L = 32 instance with three segments of very different weights.  K4h is OPT-IN (hm_workspace_set_k4_split): the suite, the
    entry points and the bench line run the fp32-input kernel, because K4h's 1e-7 ... 7e-7 difference moved one exact
    per-iteration count of test_gpu_configs.py::test_frame_turns_invalid_mid_trajectory_L256 off the oracle's."""
    import ctypes
    from hortimapping_amd import _lib, optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    lib = _lib.lib()
    lib.hm_workspace_set_k4_split.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for L, it, kw in ((256, 1, dict(n_pts=1024, n_frames=1, n_fg=32, n_bg=32)), (32, 1, dict(n_pts=2000, n_frames=4, n_fg=120, n_bg=100))):
        p = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
        dec = DecoderWeights.from_params(p).set_precision("f16x3")
        Ws, bs = S.fold_weight_norm(p)
        fac = W.gpu_sdf_factory(dec)
        insts = [W.to_instance(S.make_instance(Ws, bs, L, i, sdf_fn_factory=fac, **kw)) for i in range(3)]
        opt = W.c2_opt_cfg(max_iter=it, n_sample_on_ray=16 if L == 256 else 30, n_frame=kw["n_frames"])
        opt["robust_iter"] = 0                                   # Huber weights on in the one iteration compared
        hcfg = HO.opt_cfg_from_dict(opt)
        out = {}
        for split in (0, 1, 2):
            pb = HO.PackedBatch(insts, L, kw["n_frames"], "cuda")
            ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray)
            assert lib.hm_workspace_set_k4_split(ws.handle, split) == 0
            dbg = {}
            HO.run_packed(ws, hcfg, pb, 0, dbg)
            torch.cuda.synchronize()
            out[split] = (dbg["A"].cpu().numpy(), dbg["b"].cpu().numpy(), pb.latent.cpu().numpy(), pb.T_ow.cpu().numpy())
            ws.release()
        for split, name in ((1, "K4h (64 x 64 tiles)"), (2, "K4w in experimental builds, K4h otherwise")):
            for b in range(3):
                A0, A1 = np.tril(out[0][0][b]), np.tril(out[split][0][b])
                # ONE iteration from identical inputs: the kernels see the same rows, so the damped systems differ by the
                # summation arithmetic only (fp32 MFMA chain vs exact fp16 products + fp32 accumulation, 2^-22 dropped term;
                # K4w also takes the square root of the row weights)
                assert np.isfinite(A1).all()
                eA = np.abs(A1 - A0).max() / np.abs(A0).max()
                eb = np.abs(out[split][1][b] - out[0][1][b]).max() / np.abs(out[0][1][b]).max()
                print(f"L={L} inst {b}: {name} vs fp32 K4: H {eA:.1e}, b {eb:.1e}")
                assert eA <= 2e-6 and eb <= 2e-6, (L, b, name, eA, eb)
                assert np.abs(out[split][2][b] - out[0][2][b]).max() <= 1e-4 * max(np.abs(out[0][2][b]).max(), 1e-3)
