"""Edges of the operator's envelope on the GPU: the maximum ray-sample count (64 = one wavefront lane per sample),
ragged frame / ray / point counts inside one batch (padding must never leak into results), and the error behaviour of
the C ABI when a request exceeds the workspace limits (status code + hm_last_error text, never a crash)."""
import os

import pytest
import torch

from golden_util import PRECISIONS, T as TOL
from test_gpu_configs import make, oracle_run, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(params=PRECISIONS, autouse=True, scope="module")
def precision(request):
    os.environ["HM_PRECISION"] = request.param
    yield request.param
    os.environ.pop("HM_PRECISION", None)


def test_max_samples_and_ragged_batch_vs_oracle():
    from hortimapping_amd import optimizer as HO, workloads as W
    opt = W.c2_opt_cfg(max_iter=4, n_sample_on_ray=64, n_frame=8)
    shapes = [dict(n_pts=130, n_frames=1, n_fg=20, n_bg=9), dict(n_pts=700, n_frames=3, n_fg=64, n_bg=31),
              dict(n_pts=65, n_frames=6, n_fg=7, n_bg=50)]
    insts, refs = [], []
    for j, kw in enumerate(shapes):
        dec, od, dicts = make(32, 4, 0.04, (1.0, 0.75, 1.3), [10 + j], **kw)
        insts.append(W.to_instance(dicts[0], pose_known=True))
        refs.append((od, dicts[0]))
    res = HO.optimize_batch(dec, opt, insts)
    singles = [HO.optimize_batch(dec, opt, [i])[0] for i in insts]
    for j, (r, s, (od, d)) in enumerate(zip(res, singles, refs)):
        assert torch.equal(r.latent, s.latent) and torch.equal(r.T_ow, s.T_ow)      # padding-independent, bit for bit
        z, T, n = oracle_run(od, opt, d, True, ("lim", j))
        assert r.iter_count == n == 4
        assert rel(r.latent, z) < TOL(2e-3, 5e-2) and rel(r.T_ow, T) < TOL(1e-4, 3e-3)


def test_requests_beyond_the_workspace_limits_fail_loudly():
    from hortimapping_amd import _lib, optimizer as HO, workloads as W
    dec, od, dicts = make(32, 4, 0.04, (1.0, 0.75, 1.3), [0, 1], n_pts=100, n_frames=1, n_fg=8, n_bg=8)
    insts = [W.to_instance(d) for d in dicts]
    dev = torch.device("cuda")
    pb = HO.PackedBatch(insts, 32, 1, dev, joint=True)
    with pytest.raises(_lib.HortiHipError, match="limits"):
        HO.Workspace(dec, 2, pb.points_stride, pb.F, pb.R, 65)                      # more samples than wavefront lanes
    ws = HO.Workspace(dec, 1, pb.points_stride, pb.F, pb.R, 16)                     # workspace sized for ONE instance
    with pytest.raises(_lib.HortiHipError, match="exceeds workspace limit"):
        HO.run_packed(ws, HO.opt_cfg_from_dict(W.c2_opt_cfg(max_iter=2)), pb, 0)
    ws2 = HO.Workspace(dec, 2, pb.points_stride, pb.F, pb.R, 16)
    with pytest.raises(_lib.HortiHipError, match="n_sample_on_ray"):
        HO.run_packed(ws2, HO.opt_cfg_from_dict(W.c2_opt_cfg(max_iter=2, n_sample_on_ray=32)), pb, 0)
    # the handles stay usable after a refused call
    HO.run_packed(ws2, HO.opt_cfg_from_dict(W.c2_opt_cfg(max_iter=2)), pb, 0)
    torch.cuda.synchronize()
    assert int(pb.iter_count.min()) == 2


def test_frame_without_enough_ball_samples_is_skipped_like_the_reference():
    """One of three frames looks past the fruit (fewer than min_valid_sample = 100 ball-valid samples): the reference's
    compute_render_loss returns None for it and the iteration goes on with the other frames (optimizer.py:130-132);
    a second instance in the same batch keeps all its frames."""
    from hortimapping_amd import optimizer as HO, workloads as W
    opt = W.c2_opt_cfg(max_iter=4, n_sample_on_ray=16, n_frame=3)
    dec, od, dicts = make(32, 4, 0.04, (1.0, 0.75, 1.3), [20, 21], n_pts=400, n_frames=3, n_fg=60, n_bg=40)
    off = dicts[0]["render"]["T_wc"][1].copy()
    off[0, 3] += 0.6                                    # camera 1 of instance 0 moved 60 cm sideways: rays miss the ball
    dicts[0]["render"]["T_wc"][1] = off
    insts = [W.to_instance(d, pose_known=True) for d in dicts]
    dbg = {}
    res = HO.optimize_batch(dec, opt, insts, debug=dbg)
    counts = dbg["counts"].cpu()                         # [K_v, K_g, V, -] of the last iteration
    full = HO.optimize_batch(dec, opt, [W.to_instance(dicts[1], pose_known=True)])[0]
    assert torch.equal(res[1].latent, full.latent)
    for j, (r, d) in enumerate(zip(res, dicts)):
        z, T, n = oracle_run(od, opt, d, True, ("skipframe", j))
        assert r.iter_count == n == 4 and r.status == (8 | 64 if j == 0 else 8)     # 64 = HM_STATUS_FRAME_SKIPPED
        assert rel(r.latent, z) < TOL(2e-3, 5e-2) and rel(r.T_ow, T) < TOL(1e-4, 3e-3)
    assert int(counts[0, 0]) < int(counts[1, 0])         # the skipped frame's samples are not counted for instance 0


def test_instance_beyond_a_capacity_is_refused_not_truncated():
    """C-ABI callers can lie about per-instance counts (the Python packer cannot): an instance whose n_points / ray
    counts exceed the packed strides, or whose Jacobian samples exceed max_grad_samples, must come back flagged
    HM_STATUS_LIMIT (128) and its neighbours in the batch must be untouched by it -- never a silently wrong H."""
    from hortimapping_amd import _lib, optimizer as HO, workloads as W
    dec, od, dicts = make(32, 4, 0.04, (1.0, 0.75, 1.3), [0, 1, 2], n_pts=100, n_frames=1, n_fg=60, n_bg=24)
    insts = [W.to_instance(d, pose_known=True) for d in dicts]
    dev = torch.device("cuda")
    cfg = HO.opt_cfg_from_dict(W.c2_opt_cfg(max_iter=3))
    good = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=3), insts)
    # (a) n_points[1] larger than the stride, n_fg[2] + n_bg[2] larger than the ray capacity
    pb = HO.PackedBatch(insts, 32, 1, dev, joint=True)
    ws = HO.Workspace(dec, 3, pb.points_stride, pb.F, pb.R, 16)
    lat0 = pb.latent.clone()
    pb.n_points[1] = pb.points_stride + 1
    pb.n_fg[2, 0] = pb.R
    HO.run_packed(ws, cfg, pb, 0)
    torch.cuda.synchronize()
    st, it = pb.status.cpu(), pb.iter_count.cpu()
    assert int(st[0]) == 8 and int(it[0]) == 3
    assert torch.equal(pb.latent[0].cpu(), good[0].latent)
    for b in (1, 2):
        assert int(st[b]) == HO.STATUS_LIMIT and int(it[b]) == 0
        assert torch.equal(pb.latent[b], lat0[b])                                   # state untouched
    # (b) a points_stride beyond the workspace's capacity is refused on the host
    ws_small = HO.Workspace(dec, 3, 64, pb.F, pb.R, 16)
    pb2 = HO.PackedBatch(insts, 32, 1, dev, joint=True)
    with pytest.raises(_lib.HortiHipError, match="points_stride"):
        HO.run_packed(ws_small, cfg, pb2, 0)
    # (c) fewer Jacobian-sample slots than surviving samples: reported, instance stopped
    ws_cap = HO.Workspace(dec, 3, pb2.points_stride, pb2.F, pb2.R, 16, max_grad_samples=64)
    HO.run_packed(ws_cap, cfg, pb2, 0)
    torch.cuda.synchronize()
    st = pb2.status.cpu()
    assert all(int(s) & HO.STATUS_LIMIT for s in st) and all(int(s) & HO.STATUS_SOLVE_FAILED for s in st)
    assert int(pb2.iter_count.max()) == 0
