"""Full-size records of the ACTUAL reference loop for BASELINE.json configs[1] (C2-joint, L=256, 200 LM iterations).

Build-container only (needs the read-only mount /root/reference; `oracle/ref_shim.py` imports it on CPU).  Runs the
reference's own `Optimizer.shape_pose_joint_opt` (`wild_completion/optimizer.py:28-302`) -- its autograd Jacobians, its
`torch.inverse`, its loss builders, nothing of ours in the loop -- on instances of `tests/golden/c2_fullsize_inputs.npz`
(the very inputs the GPU box optimises), in both pose modes, on the nominal inputs and on the four structured
1e-7-relative perturbations of `make_fullsize_records.py` (points x(1+-1e-7), initial pose x(1+1e-7), foreground depths
x(1+1e-7); the same `perturb()` is imported, so reference and oracle see bit-identical perturbed inputs), and commits

    tests/golden/c2_fullsize_reference.npz      latent / T_ow / iter_count of every reference run

What it pins (tests/test_fullsize_reference_cpu.py on CPU, tests/test_gpu_fullsize.py on the GPU box):
  * the oracle's 200-iteration result deviates from the reference's by no more than the reference deviates from ITSELF
    under a one-ulp input change, instance by instance;
  * the noise_i that calibrates the GPU gate is the reference's, not only the oracle's.

Usage:  python tests/golden/make_reference_records.py [--instances 0-15] [--workers 6] [--iters 200] [--case c2|wc]
One run = 200 iterations x ~1.5 s on one thread (5 min); runs are independent processes with one thread each; partial
results live under /tmp/c2_reference_records so the script can be resumed / extended.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

L = 256
PERTS = ("nominal", "points_up", "points_down", "pose0_up", "depth_up")
MODES = ("known", "free")
CASES = {
    # case -> (inputs fixture, output fixture, cfg keyword arguments of workloads.c2_opt_cfg)
    "c2": ("c2_fullsize_inputs.npz", "c2_fullsize_reference.npz", dict()),
    "wc": ("wc_fullsize_inputs.npz", "wc_fullsize_reference.npz", dict()),
    # the C2 workload on the TRAINED decoder (dense layers; tests/golden/trained_decoder_L256.npz)
    "trained": ("trained_c2_inputs.npz", "trained_c2_reference.npz", dict()),
}


def _perturb():
    argv, sys.argv = sys.argv, sys.argv[:1]           # make_fullsize_records reads positional argv at import
    try:
        import make_fullsize_records as MF
    finally:
        sys.argv = argv
    return MF.perturb


def load_instance(inp, i):
    """instance i of an inputs fixture as the dict layout `perturb()` expects (frames as lists)."""
    r = {}
    for k in ("T_wc", "rays_fg", "rays_bg", "depth_fg", "depth_bg"):
        a = inp[k][i]
        # c2 fixture: one frame per instance, stored without a frame axis; multi-frame fixtures carry `n_frames`
        r[k] = [a[f].copy() for f in range(int(inp["n_frames"][i]))] if "n_frames" in inp.files else [a.copy()]
    return {"latent0": inp["latent0"][i].copy(), "T_ow0": inp["T_ow0"][i].copy(), "points_w": inp["points_w"][i].copy(),
            "render": r, "cube_radius": float(inp["cube_radius"][i])}


_REF = None


def _run(task):
    case, i, mode, pert, n_iter, scratch = task
    out = os.path.join(scratch, f"{case}_{i:03d}_{mode}_{pert}_{n_iter}.npz")
    if os.path.exists(out):
        return out
    import torch
    torch.set_num_threads(1)
    global _REF
    if _REF is None:
        from oracle import ref_shim
        from hortimapping_amd import synthetic as S
        ns = ref_shim.import_reference()
        from hortimapping_amd import workloads as W0
        if case == "trained":
            with np.load(os.path.join(HERE, "trained_decoder_L256.npz")) as f:
                p = {k: (int(f[k]) if k in ("latent_dim", "hidden") else f[k]) for k in f.files if k.startswith("lin") or k in ("latent_dim", "hidden")}
        else:
            p = W0.wc_decoder_params(L) if case == "wc" else S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
        _REF = (ns, ref_shim.build_reference_decoder(ns, p), _perturb())
    ns, dec, perturb = _REF
    from hortimapping_amd import workloads as W
    inp = np.load(os.path.join(HERE, CASES[case][0]))
    d = perturb(load_instance(inp, i), pert)
    cfg = {"device": "cpu", "opt": (W.wc_opt_cfg(max_iter=n_iter) if case == "wc" else W.c2_opt_cfg(max_iter=n_iter)),
           "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}}
    opt = ns.optimizer.Optimizer(cfg, dec, None, None)
    t = torch.from_numpy
    rd = {k: [t(a) for a in v] for k, v in d["render"].items()}
    t0 = time.time()
    z, T, n = opt.shape_pose_joint_opt(t(d["latent0"]), t(d["T_ow0"]), rd, t(d["points_w"]), d["cube_radius"], None,
                                       pose_known=(mode == "known"))
    np.savez(out + ".tmp.npz", latent=z.detach().numpy(), T_ow=T.detach().numpy(), iter_count=n, seconds=time.time() - t0)
    os.replace(out + ".tmp.npz", out)
    return out


def parse_ids(s):
    ids = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            ids += list(range(int(a), int(b) + 1))
        else:
            ids.append(int(part))
    return ids


def main():
    global MODES, PERTS
    import multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", default="0-7")
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--case", default="c2", choices=sorted(CASES))
    ap.add_argument("--scratch", default="/tmp/c2_reference_records")
    ap.add_argument("--assemble-only", action="store_true")
    ap.add_argument("--modes", default="known,free")
    ap.add_argument("--perts", default="nominal,points_up,points_down,pose0_up,depth_up")
    a = ap.parse_args()
    MODES, PERTS = tuple(a.modes.split(",")), tuple(a.perts.split(","))
    ids = parse_ids(a.instances)                     # positions in the inputs fixture
    os.makedirs(a.scratch, exist_ok=True)
    tasks = [(a.case, i, m, p, a.iters, a.scratch) for i in ids for p in PERTS for m in MODES]
    if not a.assemble_only:
        ctx = mp.get_context("spawn")
        t0 = time.time()
        with ctx.Pool(a.workers) as pool:
            for k, _ in enumerate(pool.imap_unordered(_run, tasks)):
                print(f"{k + 1}/{len(tasks)} reference runs, {time.time() - t0:.0f} s", flush=True)
    rec = {}
    for m in MODES:
        lat = np.zeros((len(PERTS), len(ids), L), np.float32)
        Tow = np.zeros((len(PERTS), len(ids), 4, 4), np.float32)
        itc = np.zeros((len(PERTS), len(ids)), np.int32)
        for pi, p in enumerate(PERTS):
            for k, i in enumerate(ids):
                r = np.load(os.path.join(a.scratch, f"{a.case}_{i:03d}_{m}_{p}_{a.iters}.npz"))
                lat[pi, k], Tow[pi, k], itc[pi, k] = r["latent"], r["T_ow"], r["iter_count"]
        rec[f"{m}_latent"], rec[f"{m}_T_ow"], rec[f"{m}_iter_count"] = lat, Tow, itc
    name = CASES[a.case][1] if a.iters == 200 else CASES[a.case][1].replace(".npz", f"_it{a.iters}.npz")
    np.savez_compressed(os.path.join(HERE, name), inst_ids=np.array(ids, np.int32), perts=np.array(PERTS),
                        n_iter=a.iters, eps=1e-7, **rec)
    print("written", name, flush=True)


if __name__ == "__main__":
    main()
