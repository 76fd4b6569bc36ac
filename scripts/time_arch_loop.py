#!/usr/bin/env python3
"""The full C2-joint job (64 peppers, L = 256, 200 forced LM iterations) in EXACT fp32 on the specialised decoder kernel
and on the any-architecture kernel (the shipped layer table forced through hm_decoder_create_arch), plus a job on a
table the specialised kernels cannot run (4 x 256, latent_in = [2], L = 64, analytic fruit).  Shows that
hm_optimize_batch runs whole jobs on such handles, what that costs, and that the two f32 kernels agree on the outcome.
GPU box:  python scripts/time_arch_loop.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("HM_PRECISION", None)
from hortimapping_amd import optimizer as HO, synthetic as S, workloads as W     # noqa: E402
from hortimapping_amd.decoder import DecoderWeights                              # noqa: E402


def run(dec, cfg, insts, reps=2):
    cache = {}
    HO.optimize_batch(dec, cfg, insts, cache=cache)                # warm-up: workspace allocation, code load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = HO.optimize_batch(dec, cfg, insts, cache=cache)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, res


def main():
    p = S.make_synthetic_decoder(256, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
    fix = DecoderWeights.from_params(p)
    gen = DecoderWeights.from_params(p, force_generic=True)
    dicts = W.make_c2_instances(p, fix, range(64), kind="joint")
    insts = [W.to_instance(d) for d in dicts]
    cfg = W.c2_opt_cfg()
    rows = []
    t_fix, r_fix = run(fix, cfg, insts)
    t_gen, r_gen = run(gen, cfg, insts)
    dz = max(float((a.latent - b.latent).abs().max() / b.latent.abs().max()) for a, b in zip(r_gen, r_fix))
    its = sorted({r.iter_count for r in r_gen} | {r.iter_count for r in r_fix})
    rows.append(("C2-joint, shipped table, k_decoder (exact f32)", 64 / t_fix, t_fix))
    rows.append(("C2-joint, shipped table, k_decoder_any (exact f32)", 64 / t_gen, t_gen))
    t_fh, _ = run(fix.set_precision("f16x3"), cfg, insts)
    t_gh, r_gh = run(gen.set_precision("f16x3"), cfg, insts)
    rows.append(("C2-joint, shipped table, k_decoder_h (f16x3, fused render chain)", 64 / t_fh, t_fh))
    rows.append(("C2-joint, shipped table, k_decoder_any_h (f16x3, split render chain)", 64 / t_gh, t_gh))
    # a table only the any-architecture kernel runs
    pa = S.make_arch_decoder(64, [256] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], weight_norm=True, seed=31, analytic=True)
    da = DecoderWeights.from_params(pa)
    fac = W.gpu_sdf_factory(da)
    da_d = [S.make_instance(None, None, 64, i, n_pts=1024, n_frames=1, n_fg=32, n_bg=32, sdf_fn_factory=fac) for i in range(64)]
    ia = [W.to_instance(d) for d in da_d]
    t_a, r_a = run(da, cfg, ia)
    rows.append(("C2-joint sizes, 4 x 256 / latent_in [2] / L = 64, k_decoder_any (f32)", 64 / t_a, t_a))
    t_ah, _ = run(da.set_precision("f16x3"), cfg, ia)
    rows.append(("C2-joint sizes, 4 x 256 / latent_in [2] / L = 64, k_decoder_any_h (f16x3)", 64 / t_ah, t_ah))
    print("64 instances x 200 forced LM iterations, one MI355X")
    for label, ips, t in rows:
        print("%-78s %7.1f instances/s  (%.3f s per job)" % (label, ips, t))
    print("k_decoder_any vs k_decoder after 200 iterations: iteration counts %s, max relative latent difference %.2e "
          "(two exact-f32 kernels with different summation orders on a chaotic loop)" % (its, dz))
    st = sorted({r.status for r in r_a})
    print("4 x 256 job: iteration counts %s, status words %s, mean |z| %.4f" %
          (sorted({r.iter_count for r in r_a}), st, float(np.mean([float(r.latent.abs().mean()) for r in r_a]))))


if __name__ == "__main__":
    main()
