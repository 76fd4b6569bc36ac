"""Round-6 GPU tests (run with `-m gpu`): concurrent (decoder, YAML) groups of BASELINE configs[4] and the leased group
streams behind them."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(name, max_iter):
    o = yaml.safe_load(open(os.path.join(ROOT, "configs", name)))["opt"]
    o["converge"]["max_iter"] = max_iter
    return o


def _group(seed, r0, aniso, opt, n, **shape):
    from hortimapping_amd import synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    p = S.make_synthetic_decoder(32, seed=seed, r0=r0, aniso=aniso)
    dec = DecoderWeights.from_params(p).set_precision("f16x3")
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)               # ray-marches the synthetic fruit on the GPU (the numpy forward takes 20 s per fruit)
    protos = [S.make_instance(Ws, bs, 32, i, sdf_fn_factory=fac, **shape) for i in range(4)]
    return dec, opt, [W.to_instance(protos[i % 4], pose_known=False) for i in range(n)]


def _same(a, b):
    return (a.iter_count == b.iter_count and a.status == b.status and torch.equal(a.latent, b.latent)
            and torch.equal(a.T_ow, b.T_ow))


def test_concurrent_groups_give_the_bits_of_back_to_back_calls():
    """`optimize_grouped` runs the pepper and the berry group of configs[4] at the same time (one host thread, one side
    stream and a lease of group streams per call): every result equals the result of the same group run alone, bit for
    bit, in job order; early exits on (the shipped thresholds), 20 instances per group so that each call forks into two
    instance groups of its own (four leased streams in flight)."""
    from hortimapping_amd import optimizer as HO
    dec_p, opt_p, ip = _group(1, 0.04, (1.0, 0.75, 1.3), _opt("lab_pepper.yaml", 12), 20, n_pts=600, n_frames=3, n_fg=120,
                              n_bg=60)
    dec_b, opt_b, ib = _group(3, 0.02, (1.0, 1.2, 0.9), _opt("lab_berry.yaml", 12), 20, n_pts=500, n_frames=3, n_fg=160,
                              n_bg=80, r_max=0.04)
    jobs = []
    for a, b in zip(ip, ib):                                   # interleaved job list
        jobs += [(dec_p, opt_p, a), (dec_b, opt_b, b)]
    alone_p = HO.optimize_batch(dec_p, opt_p, ip)
    alone_b = HO.optimize_batch(dec_b, opt_b, ib)
    for rep in range(3):                                       # scheduling differs from run to run: the bits must not
        res = HO.optimize_grouped(jobs)
        assert all(_same(r, e) for r, e in zip(res[0::2], alone_p)), rep
        assert all(_same(r, e) for r, e in zip(res[1::2], alone_b)), rep
    serial = HO.optimize_grouped(jobs, concurrent=False)
    assert all(_same(r, e) for r, e in zip(serial, res))
    assert len({r.iter_count for r in res}) > 1                # ragged exits: the case the concurrency is for


def test_run_concurrent_propagates_errors_and_orders_results():
    from hortimapping_amd import optimizer as HO

    def boom():
        raise ValueError("from a worker thread")
    with pytest.raises(ValueError):
        HO.run_concurrent([lambda: 1, boom])
    x = torch.arange(8, device="cuda", dtype=torch.float32)
    out = HO.run_concurrent([lambda: (x * 2).sum().item(), lambda: (x + 1).sum().item(), lambda: 3])
    assert out == [56.0, 36.0, 3]


def test_a_call_that_finds_the_pool_leased_still_gives_the_same_bits():
    """Three calls in flight at once on three workspaces: the pool has four streams, so the third call gets no lease of
    two and runs as one group on its own stream -- the result must not know."""
    from hortimapping_amd import optimizer as HO
    dec, opt, insts = _group(1, 0.04, (1.0, 0.75, 1.3), _opt("lab_pepper.yaml", 6), 16, n_pts=400, n_frames=2, n_fg=80,
                             n_bg=40)
    alone = HO.optimize_batch(dec, opt, insts)
    res = HO.run_concurrent([(lambda: HO.optimize_batch(dec, opt, insts)) for _ in range(3)])
    for r in res:
        assert all(_same(a, b) for a, b in zip(r, alone))
