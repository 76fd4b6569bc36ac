#!/usr/bin/env python3
"""Timing of the any-architecture decoder kernel (csrc/hm_decoder_any.hip) next to the specialised exact-f32 kernel on
the SHIPPED layer table (same weights, same queries), plus a few other tables.  GPU box:  python scripts/time_arch_decoder.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hortimapping_amd import ops, synthetic as S                     # noqa: E402
from hortimapping_amd.decoder import DecoderWeights                   # noqa: E402


def flops_per_query(table, mode):
    f = sum(2 * i * o for i, o in zip(table["in_dim"], table["out_dim"]))
    return f * (2 if mode == 1 else 1)


def timeit(dec, B, n, mode, reps=20):
    L = dec.latent_dim
    lat = (0.07 * torch.randn(B, L)).cuda()
    pts4 = torch.zeros(B, n, 4)
    pts4[..., :3] = 0.04 * torch.randn(B, n, 3)
    pts4 = pts4.cuda()
    nq = torch.full((B,), n, dtype=torch.int32).cuda()
    for _ in range(3):
        ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    dev = sorted(a.elapsed_time(b) * 1e-3 for a, b in ev)
    return dev[len(dev) // 2], wall, dev[0], dev[-1]


def main():
    os.environ.pop("HM_PRECISION", None)
    B, n = 64, 1024
    p = S.make_synthetic_decoder(256, seed=2)
    rows = []
    for label, dec in (("shipped table, specialised f32 kernel", DecoderWeights.from_params(p)),
                       ("shipped table, any-architecture kernel f32", DecoderWeights.from_params(p, force_generic=True)),
                       ("shipped table, specialised f16x3 kernel", DecoderWeights.from_params(p).set_precision("f16x3")),
                       ("shipped table, any-architecture f16x3", DecoderWeights.from_params(p, force_generic=True).set_precision("f16x3"))):
        for mode in (0, 1):
            t, wall, lo, hi = timeit(dec, B, n, mode)
            rows.append((label, mode, t, flops_per_query(dec.table, mode) * B * n / t / 1e12, wall, lo, hi))
    for label, kw in (("4 x 256, latent_in [2], weight norm", dict(latent_dim=64, dims=[256] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=True)),
                      ("8 x 512, latent_in [4], LayerNorm", dict(latent_dim=256, dims=[512] * 8, latent_in=[4], norm_layers=list(range(8)), weight_norm=False)),
                      ("6 x 128 plain", dict(latent_dim=32, dims=[128] * 6))):
        dec = DecoderWeights.from_params(S.make_arch_decoder(seed=1, **kw))
        for arith in ("f32", "f16x3"):
            dec.set_precision(arith)
            for mode in (0, 1):
                t, wall, lo, hi = timeit(dec, B, n, mode)
                rows.append((label + ", " + arith, mode, t, flops_per_query(dec.table, mode) * B * n / t / 1e12, wall, lo, hi))
    print("decode API, %d instances x %d queries (pose_dim 7); TFLOP/s on the dense layer-table flop" % (B, n))
    print("(median of 20 calls by device events, memset of the output buffers included; [min .. max]; host wall per call)")
    for label, mode, t, tf, wall, lo, hi in rows:
        print("%-48s %-8s %8.3f ms  %7.1f TFLOP/s   [%.3f .. %.3f]  wall %.3f" %
              (label, "fwd" if mode == 0 else "fwd+bwd", t * 1e3, tf, lo * 1e3, hi * 1e3, wall * 1e3))


if __name__ == "__main__":
    main()
