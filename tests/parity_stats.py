"""Distributional parity statistics for the full-size (200-iteration) gates.

The 200-iteration LM map of the reference is chaotic where the problem is ill conditioned (free pose at the C2 size): a
1e-7 relative change of an input moves the final Chamfer distance by up to several percent.  There a per-instance bound
"GPU within k x the largest of K perturbed runs" is a weak test with heavy tails (round 2 needed an outlier allowance).
What CAN be tested sharply is the statement "the candidate's deviation from the nominal run is ONE MORE DRAW from the
distribution of the perturbed runs' deviations": per instance the RANK of the candidate's deviation among the K
perturbed deviations is then uniform on {0..K}, and over n independent instances the normalised ranks are an i.i.d.
uniform sample -- testable with a one-sided Kolmogorov-Smirnov statistic (alternative: the candidate's deviations are
stochastically LARGER, i.e. it is worse than a 1-ulp input change)."""
import numpy as np


def rank_fraction(d, D):
    """d: (n,) candidate deviations; D: (K, n) deviations of the K perturbed runs.  Returns u in [0, 1]: the fraction of
    the K perturbed deviations below the candidate's (ties count half).  Uniform on {0, 1/K, .., 1} under H0."""
    d = np.asarray(d, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    return ((D < d[None]).sum(axis=0) + 0.5 * (D == d[None]).sum(axis=0)) / D.shape[0]


def ks_upper(u, K):
    """One-sided KS distance of the normalised ranks `u` (values j/K) from the discrete uniform law on {0, 1/K, .., 1},
    towards LARGER ranks: D+ = max_x [F0(x) - F_emp(x)], evaluated at the K+1 support points.  Returns (D+, p) with the
    asymptotic one-sided tail p = exp(-2 n D+^2) (conservative for a discrete null)."""
    u = np.asarray(u, dtype=np.float64)
    n = len(u)
    if n == 0:
        return 0.0, 1.0
    xs = np.arange(K + 1) / K
    f0 = (np.arange(K + 1) + 1.0) / (K + 1)                  # P(U <= x) under the discrete uniform
    femp = np.array([(u <= x + 1e-12).mean() for x in xs])
    dplus = float(max(0.0, (f0 - femp).max()))
    return dplus, float(np.exp(-2.0 * n * dplus * dplus))


def ks_min_p(n, K):
    """Smallest p the KS gate can return with n ranked instances and K perturbed runs (every candidate beyond all of its
    perturbed runs: D+ = K / (K + 1)).  A gate whose min p is not below alpha can never fail: it has no power."""
    if n == 0:
        return 1.0
    d = K / (K + 1.0)
    return float(np.exp(-2.0 * n * d * d))


def exchange_test(dev, pert_dev, n_mc=20000, seed=0):
    """Exchangeability test with magnitudes, over ALL instances (no floor, no ranking of a few survivors):
        T = mean_i log( dev_i / max_k pert_dev[k, i] )
    against its null distribution "the candidate is one more perturbed run": for every instance draw one of the K
    perturbed runs as a stand-in candidate and compare it with the max of the OTHER K - 1 (a max over one run fewer is a
    little smaller, which makes the stand-ins look slightly worse, i.e. the test slightly conservative).  One-sided
    Monte-Carlo p = P(T_null >= T).  Instances where every deviation is exactly zero (a metric the mode pins, e.g. the
    rotation in pose_known mode) carry no information and are dropped.  Where the rank / KS gate keeps 5-10 instances in
    pose_known mode and cannot tell 1e-3 Jacobians from fp32-class ones (round-3 VERDICT), this statistic uses all 64
    and separates them by many sigma (mean log-ratio -1.4 for f32 / f16x3 against -0.3 for the mixed mode)."""
    dev = np.asarray(dev, dtype=np.float64)
    D = np.asarray(pert_dev, dtype=np.float64)
    K, n = D.shape
    live = (D.max(axis=0) > 0) | (dev > 0)
    dev, D = dev[live], D[:, live]
    n = dev.shape[0]
    if n == 0 or K < 2:
        return {"n": n, "T": 0.0, "null_mean": 0.0, "null_sd": 0.0, "p": 1.0, "z": 0.0}
    tiny = 1e-300
    T = float(np.mean(np.log(np.maximum(dev, tiny) / np.maximum(D.max(axis=0), tiny))))
    # leave-one-out ratios of the perturbed runs themselves: r[k, i] = D[k, i] / max_{j != k} D[j, i]
    order = np.sort(D, axis=0)
    top, second = order[-1], order[-2]
    others_max = np.where(D == top[None], second[None], top[None])      # (ties: the max itself)
    r = np.log(np.maximum(D, tiny) / np.maximum(others_max, tiny))
    rs = np.random.RandomState(seed)
    pick = rs.randint(0, K, size=(n_mc, n))
    Tn = r[pick, np.arange(n)[None]].mean(axis=1)
    p = float((np.sum(Tn >= T) + 1.0) / (n_mc + 1.0))
    sd = float(Tn.std())
    return {"n": n, "T": T, "null_mean": float(Tn.mean()), "null_sd": sd, "p": p, "z": float((T - Tn.mean()) / max(sd, 1e-30))}


def gate(dev, pert_dev, floor, alpha=1e-3, selection="symmetric"):
    """Full-size gate for ONE metric.  dev: (n,) |m_candidate - m_nominal|; pert_dev: (K, n) |m_pert_k - m_nominal|;
    floor: (n,) the outright tolerance (BASELINE.json: 1e-4 relative).  An instance inside the floor passes outright
    (reported as `outright`); the rank test runs over the instances on which the comparison is informative.

    WHICH instances are ranked matters (round 5).  Rounds 3-4 ranked exactly the instances whose
    CANDIDATE deviation exceeded the floor (`selection="candidate"`).  That conditions on the candidate being large and not
    on its K stand-ins: under the null "the candidate is one more perturbed run" the surviving ranks are then NOT uniform
    but tilted towards 1 -- the more so the more instances the floor removes.  It is what produced the "high-side rank bias"
    of the scale error (2/3 of the instances inside the floor: mean rank 0.66-0.72, p down to 0.004) for the HIP results AND,
    identically, for every one-operation variant of the CPU oracle ranked against the nominal oracle
    (profiles/r05_rank_bias_table*.txt: fp64 solve 0.72 / p 0.003, 64-row tile summation 0.71 / 0.006, ...).
    `selection="symmetric"` (the default now) keeps an instance iff the LARGEST of all K + 1 deviations -- candidate and
    stand-ins alike -- exceeds the floor: a rule that is invariant under exchanging the candidate with a stand-in, so the
    ranks of the kept instances are exactly uniform under the null (tests/test_fullsize_reference_cpu.py shows both facts on
    synthetic exchangeable data).  Returns the counts, the KS statistic / p-value and `ok`."""
    dev = np.asarray(dev, dtype=np.float64)
    pert_dev = np.asarray(pert_dev, dtype=np.float64)
    outright = dev <= floor
    if selection == "candidate":
        idx = np.nonzero(~outright)[0]
    else:
        idx = np.nonzero(np.maximum(dev, pert_dev.max(axis=0)) > floor)[0]
    K = pert_dev.shape[0]
    u = rank_fraction(dev[idx], pert_dev[:, idx]) if len(idx) else np.zeros(0)
    dplus, p = ks_upper(u, K)
    return {"n": len(dev), "outright": int(outright.sum()), "ranked": len(idx), "mean_rank": float(u.mean()) if len(idx) else 0.5,
            "top_rank": int((u >= 1.0).sum()), "ks": dplus, "p": p, "ok": bool(p >= alpha), "u": u, "idx": idx,
            "min_p": ks_min_p(len(idx), K), "has_power": bool(ks_min_p(len(idx), K) < alpha), "selection": selection}
