"""Bit equality of the experimental f16x3 decoder schedules (k_decoder_g: ping-pong wave groups = variant 1, lockstep with
raw barriers + primed four-set weight ring = variant 2; four-set ring inside k_decoder_h = tune bit 3) against the product
kernel k_decoder_h (variant 0), through the decode API and through a whole joint optimisation.  Needs an experimental
build of the library:  HORTIHIP_LIB=hortimapping_amd/variants/libhortihip_exp.so python scripts/check_k1g_bits.py"""
import sys
sys.path.insert(0, '.')
import torch
from hortimapping_amd import ops, synthetic as S, optimizer as HO, workloads as W, _lib
from hortimapping_amd.decoder import DecoderWeights

lib = _lib.lib()
assert hasattr(lib, "hm_debug_k1h_variant"), "not an experimental build (HORTIHIP_LIB)"
CASES = [(0, 0), (1, 0), (1, 1), (1, 2), (2, 0), (0, 8), (0, 0x10), (0, 0x30)]          # (variant, tune)
for L in (32, 64, 96, 128, 160, 192, 224, 256):
    dec = DecoderWeights.from_params(S.make_synthetic_decoder(L, seed=7, r0=0.04, aniso=(1.0, 0.75, 1.3), bias_sigma=0.02))
    dec.set_precision("f16x3")
    g = torch.Generator().manual_seed(L)
    B, n = 3, 200
    lat = (0.07 * torch.randn(B, L, generator=g)).cuda()
    pts4 = torch.zeros(B, 256, 4)
    pts4[..., :3] = 0.05 * torch.randn(B, 256, 3, generator=g)
    pts4 = pts4.cuda()
    nq = torch.tensor([n, 64, 1], dtype=torch.int32).cuda()
    out = {}
    for c in CASES:
        lib.hm_debug_k1h_variant(c[0]); lib.hm_debug_k1h_tune(c[1])
        y0, _ = ops.decode_batch(dec, lat, pts4, nq, mode=0)
        y1, J1 = ops.decode_batch(dec, lat, pts4, nq, mode=1, pose_dim=7)
        out[c] = (y0.clone(), y1.clone(), J1.clone())
    for c in CASES[1:]:
        for b, k in enumerate((n, 64, 1)):
            for a_, b_ in zip(out[CASES[0]], out[c]):
                assert torch.equal(a_[b, :k], b_[b, :k]), (L, c, b)
    assert torch.isfinite(out[CASES[0]][2][0, :n]).all()
    print(f"L={L}: decode API bits equal over {CASES}")
for L in (32, 256):
    p = S.make_synthetic_decoder(L, seed=2 if L == 256 else 1, r0=0.04, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(p).set_precision("f16x3")
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)
    dicts = [S.make_instance(Ws, bs, L, i, n_pts=300 + 37 * i, n_frames=1 + i % 2, n_fg=40, n_bg=24, sdf_fn_factory=fac) for i in range(4)]
    opt = W.c2_opt_cfg(max_iter=6, n_sample_on_ray=16, n_frame=2)
    out = {}
    for c in CASES:
        lib.hm_debug_k1h_variant(c[0]); lib.hm_debug_k1h_tune(c[1])
        out[c] = HO.optimize_batch(dec, opt, [W.to_instance(d) for d in dicts])
    for c in CASES[1:]:
        for a, b in zip(out[CASES[0]], out[c]):
            assert a.iter_count == b.iter_count == 6 and torch.equal(a.latent, b.latent) and torch.equal(a.T_ow, b.T_ow), (L, c)
    print(f"L={L}: joint optimisation (fused main launch + backward-only render pass) bits equal over {CASES}")
lib.hm_debug_k1h_variant(0); lib.hm_debug_k1h_tune(0)
print("OK")
