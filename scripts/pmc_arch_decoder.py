#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc pass over the any-architecture kernels: the SHIPPED layer table (L = 256) forced through
hm_decoder_create_arch, 64 x 1024 queries, five launches each of forward and forward+backward in exact f32 and in f16x3, with
the specialised kernels next to them.  (scripts/pmc_arch.sh runs it under rocprofv3 and condenses the counters.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("HM_PRECISION", None)
from hortimapping_amd import ops, synthetic as S                     # noqa: E402
from hortimapping_amd.decoder import DecoderWeights                   # noqa: E402

p = S.make_synthetic_decoder(256, seed=2)
B, n = 64, 1024
lat = (0.07 * torch.randn(B, 256)).cuda()
pts4 = torch.zeros(B, n, 4)
pts4[..., :3] = 0.04 * torch.randn(B, n, 3)
pts4 = pts4.cuda()
nq = torch.full((B,), n, dtype=torch.int32).cuda()
for force in (True, False):
    dec = DecoderWeights.from_params(p, force_generic=force)
    for arith in ("f32", "f16x3"):
        dec.set_precision(arith)
        for mode in (0, 1):
            for _ in range(5):
                ops.decode_batch(dec, lat, pts4, nq, mode=mode, pose_dim=7)
torch.cuda.synchronize()
print("done")
