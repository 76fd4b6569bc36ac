#!/usr/bin/env python3
"""g19: the CPU oracle's outcomes on the random small cases the GPU fuzz test runs (tests/test_gpu_fuzz.py), generated in the
build container so that the GPU box spends its time on GPU work:

    python tests/golden/make_golden_r6_fuzz.py          ->  tests/golden/g19_fuzz_oracle.npz

Seeds 3000-3079: disjoint from every range the oracle itself was fuzzed on against the live reference (0-999, 7000-7199:
profiles/r06_oracle_fuzz.txt; 2000-2059 was round 5's GPU range).  Per seed: the oracle's result on the case and on its two
+-1e-6 probes (final latent / pose, iteration count, exit branch, the last iteration's (ball-valid, Jacobian, emitted-ray)
counts) and sum_i cond_2(H_i) 2^-23 over the normal matrices of its solves (the forward-error bound of an fp32 solve; the
state tolerance of the test is max(2e-4, 3 x probe response, that bound)).  the instances (a few KB each) ARE stored too,
because building one is a numpy ray-march of the synthetic fruit that took the GPU box 2 s per case; decoder parameters and
option blocks are rebuilt from the seed (scripts/fuzz_oracle_vs_reference.py: draw_case / build_case(c, inst))."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import fuzz_oracle_vs_reference as F          # noqa: E402  (does not import the reference until asked to)
from oracle import hm_oracle as O             # noqa: E402

SEEDS = list(range(3000, 3080))
REASONS = ["grad", "code", "pose", "max_iter", "invalid"]


def run(p, cfg, inst, pose_known, eps):
    od = O.fold_decoder(p)
    t = F.t
    rd = {k: [t(a) for a in v] for k, v in inst["render"].items()}
    pw = (inst["points_w"] * np.float32(1 + eps)).astype(np.float32)
    tr, info = [], {}
    z, T, n = O.shape_pose_joint_opt(od, cfg["opt"], t(inst["latent0"].copy()), t(inst["T_ow0"].copy()), rd, t(pw),
                                     inst["cube_radius"], pose_known=pose_known, trace=tr, exit_info=info)
    last = (tr[-1].n_valid, tr[-1].n_keep, tr[-1].n_rays) if tr else (-1, -1, -1)
    cb = 0.0
    for x in tr:
        sv = np.linalg.svd(x.H.numpy().astype(np.float64), compute_uv=False)
        cb += sv[0] / max(sv[-1], 1e-300) * 2.0 ** -23
    return z.numpy(), T.numpy(), int(n), info["reason"], last, cb


def main():
    out = {"seeds": np.array(SEEDS, dtype=np.int64)}
    for seed in SEEDS:
        c = F.draw_case(seed)
        p, inst, cfg = F.build_case(c)
        out[f"in_{seed}_latent0"], out[f"in_{seed}_T_ow0"] = inst["latent0"], inst["T_ow0"]
        out[f"in_{seed}_points_w"], out[f"in_{seed}_cube_radius"] = inst["points_w"], np.float64(inst["cube_radius"])
        for k, v in inst["render"].items():
            for f, a in enumerate(v):
                out[f"in_{seed}_{k}_{f}"] = np.asarray(a)
        out[f"in_{seed}_n_frames"] = np.int64(len(inst["render"]["T_wc"]))
        for tag, eps in (("0", 0.0), ("p", 1e-6), ("m", -1e-6)):
            z, T, n, reason, last, cb = run(p, cfg, inst, c["pose_known"], eps)
            out[f"z_{seed}_{tag}"] = z.astype(np.float32)
            out[f"T_{seed}_{tag}"] = T.astype(np.float32)
            out[f"meta_{seed}_{tag}"] = np.array([n, REASONS.index(reason), *last], dtype=np.int64)
            if tag == "0":
                out[f"cond_{seed}"] = np.array(cb, dtype=np.float64)
        print(seed, c["L"], out[f"meta_{seed}_0"], f"cond*eps {float(out[f'cond_{seed}']):.1e}", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g19_fuzz_oracle.npz"), **out)


if __name__ == "__main__":
    main()
