"""GPU tests added in round 3: the f16x3 render chain's fused launch sequence (one grid for the SDF-term forward+backward
tiles and the ray samples' forward tiles; the render Jacobian pass BACKWARD-ONLY from saved ReLU masks) must give the very
bits of the round-2 sequence (separate forward launch, forward+backward over the with-grad samples) -- the work that was
removed is a recomputation, not an approximation."""
import os

import numpy as np
import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


def _split(on):
    from hortimapping_amd import _lib
    _lib.lib().hm_debug_split_render(int(on))


@pytest.fixture(autouse=True)
def _restore():
    yield
    _split(0)


@pytest.mark.parametrize("precision", ["f16x3", "f16x3f_f16b"])
def test_render_rows_fused_equals_split(precision):
    """compute_render_loss (hm_render_residuals) on the golden render cases: same residuals AND bit-identical Jacobian rows
    with the backward-only pass as with the forward+backward pass; both within the golden tolerance of the reference."""
    from hortimapping_amd import loss as HL
    from hortimapping_amd.decoder import DecoderWeights
    for case in ("wild", "lab", "berry", "wild256"):
        g = GU.load(f"g7_render_{case}")
        dec = DecoderWeights.from_params(GU.decoder_params(str(g["decoder"]))).set_precision(precision)
        z = torch.from_numpy(g["z"])
        for f in range(int(g["n_frames"])):
            args = (dec, z, torch.from_numpy(g[f"rays_{f}"]), torch.from_numpy(g[f"depth_fg_{f}"]), torch.from_numpy(g[f"depth_bg_{f}"]),
                    torch.from_numpy(g[f"T_oc_{f}"]), torch.from_numpy(g[f"sampled_depth_{f}"]))
            kw = dict(scale_on=bool(g["scale_on"]), log_occ_on=bool(g["log_occ_on"]), occupancy_th=float(g["occupancy_th"]),
                      object_bbx_radius=float(g[f"bbx_radius_{f}"]), occlusion_on=bool(g["occlusion_on"]))
            _split(0)
            a = HL.compute_render_loss(*args, **kw)
            _split(1)
            b = HL.compute_render_loss(*args, **kw)
            assert a is not None and b is not None and len(a) == len(b) == 6
            for x, y in zip(a, b):
                assert x.shape == y.shape and torch.equal(x, y), case
            if precision == "f16x3":
                assert GU.relmax(a[0].cpu().numpy().reshape(-1), g[f"res_d_{f}"]) < 2e-5
                assert GU.relmax(a[2].cpu().numpy()[:, 0], g[f"J_d_code_{f}"]) < 2e-5


@pytest.mark.parametrize("precision", ["f16x3", "f16x3f_f16b"])
@pytest.mark.parametrize("L", [32, 256])
def test_optimisation_fused_equals_split(precision, L):
    """Whole trajectories (ragged instances, several frames, early exits possible): latent, pose, iteration count and
    status of every instance identical bit for bit between the fused and the split launch sequence."""
    from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p = S.make_synthetic_decoder(L, seed=2 if L == 256 else 1, r0=0.04, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(p).set_precision(precision)
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)
    dicts = [S.make_instance(Ws, bs, L, i, n_pts=300 + 37 * i, n_frames=1 + i % 3, n_fg=40 + 5 * i, n_bg=24 + 3 * i, sdf_fn_factory=fac)
             for i in range(5)]
    opt = W.c2_opt_cfg(max_iter=7, n_sample_on_ray=16, n_frame=3)
    opt["converge"]["epsilon_g"] = 2e-3
    out = {}
    for split in (0, 1):
        _split(split)
        out[split] = HO.optimize_batch(dec, opt, [W.to_instance(d, pose_known=bool(i % 2)) for i, d in enumerate(dicts)])
    for a, b in zip(out[0], out[1]):
        assert a.iter_count == b.iter_count and a.status == b.status
        assert torch.equal(a.latent, b.latent) and torch.equal(a.T_ow, b.T_ow)
    assert all(torch.isfinite(r.latent).all() for r in out[0])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("L", [32, 256])
def test_instance_groups_on_internal_streams_give_the_same_bits(precision, L):
    """Round 4: hm_optimize_batch runs a batch of >= 16 instances as 2-4 instance groups on internal streams (one group's
    under-filled iteration tail beside another group's main launch).  Ragged batch (different point / frame / ray counts,
    early exits, a batch size that does not divide evenly): latent, pose, iteration count and status
    of every instance identical bit for bit for 1, 2, 3, 4 and automatic groups; work counters add up the same."""
    import ctypes
    from hortimapping_amd import _lib, optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p = S.make_synthetic_decoder(L, seed=2 if L == 256 else 1, r0=0.04, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(p).set_precision(precision)
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)
    n = 37
    protos = [S.make_instance(Ws, bs, L, i, n_pts=200 + 31 * i, n_frames=1 + i % 3, n_fg=30 + 4 * i, n_bg=20 + 3 * i, sdf_fn_factory=fac)
              for i in range(6)]
    insts = [W.to_instance(protos[i % 6], pose_known=bool(i % 2)) for i in range(n)]
    opt = W.c2_opt_cfg(max_iter=9, n_sample_on_ray=16, n_frame=3)
    opt["converge"]["epsilon_g"] = 2e-3
    hcfg = HO.opt_cfg_from_dict(opt)
    pb = HO.PackedBatch(insts, L, 3, "cuda")
    ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray)
    init = (pb.latent.clone(), pb.T_ow.clone())
    lib = _lib.lib()
    lib.hm_workspace_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.hm_workspace_counters_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_void_p]
    out = {}
    for groups in (1, 2, 3, 4, 0):
        ws.set_groups(groups)
        pb.latent.copy_(init[0]); pb.T_ow.copy_(init[1])
        lib.hm_workspace_counters(ws.handle, 1)
        HO.run_packed(ws, hcfg, pb, 0)
        c5 = (ctypes.c_longlong * 5)()
        lib.hm_workspace_counters_read(ws.handle, c5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.hm_workspace_counters(ws.handle, 0)
        out[groups] = (pb.latent.clone(), pb.T_ow.clone(), pb.iter_count.clone(), pb.status.clone(), list(c5))
    a = out[1]
    assert int((a[3] == 1).sum()) >= 1 and int((a[3] == 8).sum()) >= 1       # early exits and max_iter exits in one batch
    assert int(a[2].min()) < int(a[2].max())                             # ... and stopped early while others ran on
    for groups in (2, 3, 4, 0):
        b = out[groups]
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), (groups, a[2].tolist(), b[2].tolist(), a[3].tolist(), b[3].tolist())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), groups
        assert a[4] == b[4], (groups, a[4], b[4])
    ws.release()

