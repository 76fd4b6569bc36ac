"""GPU fuzz (round 5): the HIP path against the CPU oracle on the random small cases of
`scripts/fuzz_oracle_vs_reference.py` -- the generator whose 300-case run pins the ORACLE on the live reference
(profiles/r06_oracle_fuzz.txt).  Every switch of the loop is drawn at random: L 32 / 64, Sim(3) / SE(3), linear / logistic
occupancy (the linear cases run through the round-5 screening pass), occlusion, lm_on / lm_eye / Gauss-Newton, 1-4 frames some
without foreground or without background rays, 8-120 rays, M = 2 ... 30, background depths 0 / in front / behind, pose known / free,
start poses off by up to 6 cm (frames turn None), shrunken-fruit decoders (VALID frames that emit zero rays: the exit round 4 had no
test for), loose and shipped convergence thresholds.  Asserted per case, in exact f32 and in f16x3:
  * iter_count and the exit branch (status bits without the informational FRAME_SKIPPED) identical to the oracle's -- unless the
    oracle's own runs on 1e-6-perturbed inputs disagree among themselves on them (a knife-edge case: reported, not asserted);
  * emitted rays V / Jacobian samples / ball-valid samples of the last executed iteration identical (same condition);
  * final state within max(2e-4, 3 x the oracle's own response to the +-1e-6 perturbations, sum_i cond(H_i) 2^-23 over the
    oracle's solves: the forward-error bound of an fp32 solve -- VERDICT r05 weak #1: an undamped case with cond(H) ~ 7e4
    turns rounding-level differences of H and b into 1e-3 of the state without any difference of logic).
Round 6: the seeds moved to 3000-3079, disjoint from every range the oracle was fuzzed on against the live reference (0-999,
7000-7199: profiles/r06_oracle_fuzz.txt), and the oracle's outcomes come from a committed fixture
(tests/golden/g19_fuzz_oracle.npz, made by tests/golden/make_golden_r6_fuzz.py in the build container): the GPU box no longer
spends three CPU-oracle runs per case; a seed the fixture does not hold is computed here."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

REASON_BITS = {"grad": 1, "code": 2, "pose": 4, "max_iter": 8, "invalid": 16}
SEEDS = list(range(3000, 3080))
REASONS = ["grad", "code", "pose", "max_iter", "invalid"]
_FIX = {}


def _fixture():
    if not _FIX:
        path = os.path.join(ROOT, "tests", "golden", "g19_fuzz_oracle.npz")
        _FIX["npz"] = np.load(path) if os.path.exists(path) else None
    return _FIX["npz"]


_CASES = {}         # seed -> (decoder parameters, instance, config): generated once for both parametrisations
_ORACLE = {}        # (seed, eps) -> oracle outcome: the oracle does not depend on the GPU arithmetic under test, so the
                    # second parametrisation reuses the first one's runs (three oracle runs per case are most of this test's time)


def _fixture_instance(seed):
    g = _fixture()
    if g is None or f"in_{seed}_latent0" not in g.files:
        return None
    F_ = int(g[f"in_{seed}_n_frames"])
    rd = {k: [g[f"in_{seed}_{k}_{f}"] for f in range(F_)] for k in ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg")}
    return {"latent0": g[f"in_{seed}_latent0"], "T_ow0": g[f"in_{seed}_T_ow0"], "points_w": g[f"in_{seed}_points_w"],
            "cube_radius": float(g[f"in_{seed}_cube_radius"]), "render": rd}


def _oracle_cached(F, seed, p, cfg, inst, pose_known, eps=0.0):
    key = (seed, eps)
    if key not in _ORACLE:
        g, tag = _fixture(), {0.0: "0", 1e-6: "p", -1e-6: "m"}[eps]
        if g is not None and f"meta_{seed}_{tag}" in g.files:
            m = g[f"meta_{seed}_{tag}"]
            last = None if int(m[2]) < 0 else (int(m[2]), int(m[3]), int(m[4]))
            cond = float(g[f"cond_{seed}"]) if tag == "0" else 0.0
            _ORACLE[key] = (g[f"z_{seed}_{tag}"], g[f"T_{seed}_{tag}"], int(m[0]), REASONS[int(m[1])], last, cond)
        else:
            _ORACLE[key] = _oracle(F, p, cfg, inst, pose_known, eps)
    return _ORACLE[key]


def _oracle(F, p, cfg, inst, pose_known, eps=0.0):
    from oracle import hm_oracle as O
    od = O.fold_decoder(p)
    t = F.t
    rd = {k: [t(a) for a in v] for k, v in inst["render"].items()}
    pw = (inst["points_w"] * np.float32(1 + eps)).astype(np.float32)
    tr, info = [], {}
    z, T, n = O.shape_pose_joint_opt(od, cfg["opt"], t(inst["latent0"].copy()), t(inst["T_ow0"].copy()), rd, t(pw),
                                     inst["cube_radius"], pose_known=pose_known, trace=tr, exit_info=info)
    last = (tr[-1].n_valid, tr[-1].n_keep, tr[-1].n_rays) if tr else None
    cond = 0.0
    for x in tr:
        sv = np.linalg.svd(x.H.numpy().astype(np.float64), compute_uv=False)
        cond += sv[0] / max(sv[-1], 1e-300) * 2.0 ** -23
    return z.numpy(), T.numpy(), int(n), info["reason"], last, cond


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_hip_path_equals_oracle_on_random_small_cases(precision):
    import fuzz_oracle_vs_reference as F
    from hortimapping_amd import optimizer as HO, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    decs = {}
    knife, checked, worst = [], 0, 0.0
    reached = {}
    for seed in SEEDS:
        c = F.draw_case(seed)
        if seed not in _CASES:
            _CASES[seed] = F.build_case(c, _fixture_instance(seed))   # (without the fixture: numpy ray marching, 2 s per case)
        p, inst, cfg = _CASES[seed]
        key = (c["L"], c["bias_shift"])
        if key not in decs:
            decs[key] = DecoderWeights.from_params(p).set_precision(precision)
        zo, To, no, reason, last, cond = _oracle_cached(F, seed, p, cfg, inst, c["pose_known"])
        pert = [_oracle_cached(F, seed, p, cfg, inst, c["pose_known"], e) for e in (1e-6, -1e-6)]
        stable = all(q[2] == no and q[3] == reason and q[4] == last for q in pert)
        nz = max(F.rel(q[0], zo, 1e-3) for q in pert)
        nT = max(F.rel(q[1], To, 1e-30) for q in pert)
        dbg = {}
        res = HO.optimize_batch(decs[key], cfg["opt"], [W.to_instance(inst, pose_known=c["pose_known"])], debug=dbg)[0]
        reached[reason] = reached.get(reason, 0) + 1
        tag = f"seed {seed} {c}"
        if not stable:
            knife.append(seed)
            continue
        checked += 1
        assert res.iter_count == no, (tag, res.iter_count, no, reason)
        assert (res.status & ~64) == REASON_BITS[reason], (tag, res.status, reason)
        if last is not None and reason != "invalid":
            cnt = dbg["counts"][0].cpu().numpy()
            assert (int(cnt[0]), int(cnt[1]), int(cnt[2])) == last, (tag, cnt, last)
        ez, eT = F.rel(res.latent.numpy(), zo, 1e-3), F.rel(res.T_ow.numpy(), To, 1e-30)
        tz, tT = max(2e-4, 3 * nz, cond), max(2e-4, 3 * nT, cond)
        worst = max(worst, ez / tz, eT / tT)
        assert ez <= tz and eT <= tT, (tag, ez, nz, eT, nT, cond)
    print(f"{precision}: {checked} of {len(SEEDS)} cases asserted, knife-edge (oracle unstable under 1e-6): {knife}, exits {reached}, "
          f"largest fraction of the state tolerance used {worst:.2f}")
    assert checked >= 0.8 * len(SEEDS)
    assert reached.get("invalid", 0) >= 5 and len(reached) >= 4
