"""GPU tests of precision 3, the plain fp16 MFMA decoder (hm_decoder_p.hip; BASELINE.json configs[4] "fp16 MFMA
decoder").  It is NOT an fp32-class arithmetic: the tolerances here are fp16-class (values ~3e-3, Jacobians ~1e-2 of
the largest entry) and are stated as such; what is held exactly is everything that is not arithmetic -- tile tails of the
128-query tiling, untouched padding rows, batch == single bit for bit, determinism, the range guard."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dec(L, seed=11, **kw):
    from hortimapping_amd import synthetic as S
    from hortimapping_amd.decoder import DecoderWeights
    from oracle import hm_oracle as O
    p = S.make_synthetic_decoder(L, seed=seed, aniso=(1.0, 0.75, 1.3), wn_perturb=0.05, **kw)
    return p, DecoderWeights.from_params(p).set_precision("f16"), O.fold_decoder(p).to(torch.float64)


@pytest.mark.parametrize("L", [32, 128, 256])
def test_f16_decoder_vs_fp64_oracle_ragged_tiles(L):
    from hortimapping_amd import ops
    from oracle import hm_oracle as O
    p, dec, od = _dec(L)
    assert dec.precision == "f16"
    gen = torch.Generator().manual_seed(L)
    nq = [1, 63, 64, 65, 127, 128, 129, 200, 320, 0]       # tails on both sides of the 128-query tile and of 64
    B, N = len(nq), 320
    lat = 0.07 * torch.randn(B, L, generator=gen)
    pts = 0.04 * torch.randn(B, N, 3, generator=gen)
    pts4 = torch.zeros(B, N, 4)
    pts4[..., :3] = pts
    y0, _ = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), torch.tensor(nq, dtype=torch.int32).cuda(), mode=0)
    for pose_dim in (0, 7):
        y, J = ops.decode_batch(dec, lat.cuda(), pts4.cuda(), torch.tensor(nq, dtype=torch.int32).cuda(), mode=1,
                                pose_dim=pose_dim)
        assert torch.equal(y, y0)                            # forward-only and forward+backward kernels agree exactly
        y, J = y.cpu().double(), J.cpu().double()
        for b, k in enumerate(nq):
            if k == 0:
                assert float(J[b].abs().max()) == 0.0 and float(y[b].abs().max()) == 0.0
                continue
            yo, go = O.decoder_jacobian(od, lat[b], pts[b, :k])
            assert float((y[b, :k] - yo).abs().max()) < 3e-3 * 0.04 * 10        # sdf ~ r0 = 0.04: 1.2e-3 absolute
            assert float((J[b, :k, :L] - go[:, :L]).abs().max()) < 2e-2 * float(go[:, :L].abs().max())
            gx = J[b, :k, L:L + 3] if pose_dim == 0 else J[b, :k, L:L + 3]
            assert float((gx - go[:, L:]).abs().max()) < 2e-2 * float(go[:, L:].abs().max())
            assert float((J[b, :k, L + 7] - y[b, :k]).abs().max()) == 0.0      # residual column = sdf
            if k < N:
                assert float(J[b, k:].abs().max()) == 0.0 and float(y[b, k:].abs().max()) == 0.0   # padding untouched


def test_f16_optimisation_runs_and_is_batch_invariant():
    """A short joint optimisation in fp16 arithmetic: finite, full iteration count, within a few percent of the CPU
    oracle's state (fp16 noise through 5 LM iterations), batched == single bit for bit, run-to-run deterministic."""
    from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W
    from oracle import hm_oracle as O
    p, dec, _ = _dec(32, seed=1, r0=0.04)
    od = O.fold_decoder(p)
    Ws, bs = S.fold_weight_norm(p)
    dicts = [S.make_instance(Ws, bs, 32, i, n_pts=300, n_frames=1, n_fg=48, n_bg=48) for i in (0, 1, 2)]
    dicts[1]["points_w"] = dicts[1]["points_w"][:131]
    opt = W.c2_opt_cfg(max_iter=5)
    insts = [W.to_instance(d, pose_known=True) for d in dicts]
    res = HO.optimize_batch(dec, opt, insts)
    res2 = HO.optimize_batch(dec, opt, insts)
    for d, r, r2, inst in zip(dicts, res, res2, insts):
        assert r.iter_count == 5 and r.status == 8 and torch.isfinite(r.latent).all() and torch.isfinite(r.T_ow).all()
        assert torch.equal(r.latent, r2.latent) and torch.equal(r.T_ow, r2.T_ow)
        s = HO.optimize_batch(dec, opt, [inst])[0]
        assert torch.equal(r.latent, s.latent) and torch.equal(r.T_ow, s.T_ow)
        rd = {k: [torch.from_numpy(a) for a in v] for k, v in d["render"].items()}
        z, T, n = O.shape_pose_joint_opt(od, opt, torch.from_numpy(d["latent0"]), torch.from_numpy(d["T_ow0"]), rd,
                                         torch.from_numpy(d["points_w"]), d["cube_radius"], pose_known=True)
        assert float((r.latent - z).abs().max() / z.abs().max()) < 0.2          # fp16-class: loose on purpose
        assert float((r.T_ow - T).abs().max() / T.abs().max()) < 2e-2


def test_f16_activation_overflow_is_reported():
    from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p = S.make_synthetic_decoder(32, seed=3, r0=0.04, aniso=(1.0, 0.75, 1.3))
    big = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in p.items()}
    big["lin1.weight_g"] = big["lin1.weight_g"] * 4.0e6
    big["lin2.weight_g"] = big["lin2.weight_g"] / 4.0e6
    Ws, bs = S.fold_weight_norm(p)
    d = S.make_instance(Ws, bs, 32, 1, n_pts=256, n_frames=1, n_fg=16, n_bg=16)
    inst = W.to_instance(d, pose_known=True)
    dec = DecoderWeights.from_params(big).set_precision("f16")
    out = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=3), [inst])[0]
    assert out.status & 32 and out.iter_count < 3 and torch.equal(out.latent, inst.latent)
