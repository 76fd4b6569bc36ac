"""Full-size records of the SHAPE-ONLY loop (`Optimizer.shape_opt_deepsdf`, wild_completion/optimizer.py:306-429) -- the
`c2_sdf` line of bench.py -- at L = 256, 200 forced iterations, on instances of tests/golden/c2_fullsize_inputs.npz (their
1024 surface points and initial pose; the render data is not used by this loop):

    tests/golden/c2_sdf_fullsize_records.npz
        ref_latent [3][8][256]   the ACTUAL reference (oracle/ref_shim.py) on instances 0..7: nominal, points x(1+-1e-7)
        orc_latent [5][16][256]  the CPU oracle on instances 0..15: nominal + the four structured perturbations
                                 (pose0_up perturbs the fixed pose; depth_up is a no-op for this loop and must reproduce
                                 the nominal run bit for bit)
Build container only.  ~8 x 3 reference runs of ~6 min and 80 oracle runs of ~20 s, one thread each.

`python tests/golden/make_sdf_records.py 2048` writes the same records for the instances of the `c2_sdf` BENCH workload
(`workloads.make_c2_instances(kind="sdf")`: 2048 surface points per instance, the literal "2048 pts/instance" of
BASELINE.json) -> c2_sdf2048_inputs.npz (points_w, latent0, T_ow0 of 16 instances) + c2_sdf2048_records.npz."""
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
NPTS = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SCRATCH = "/tmp/c2_sdf_records" if NPTS == 1024 else f"/tmp/c2_sdf{NPTS}_records"
OUT = "c2_sdf_fullsize_records.npz" if NPTS == 1024 else f"c2_sdf{NPTS}_records.npz"
INPUTS = f"c2_sdf{NPTS}_inputs.npz"
REF_PERTS = ("nominal", "points_up", "points_down")
ORC_PERTS = ("nominal", "points_up", "points_down", "pose0_up", "depth_up")
_S = {}


def _setup(kind):
    if kind in _S:
        return _S[kind]
    import torch
    torch.set_num_threads(1)
    argv, sys.argv = sys.argv, sys.argv[:1]
    import make_fullsize_records as MF
    import make_reference_records as MR
    sys.argv = argv
    from hortimapping_amd import synthetic as S, workloads as W
    p = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
    if NPTS == 1024:
        inp = np.load(os.path.join(HERE, "c2_fullsize_inputs.npz"))
        load = MR.load_instance
    else:
        inp = np.load(os.path.join(HERE, INPUTS))
        load = lambda f, i: {"points_w": f["points_w"][i], "latent0": f["latent0"][i], "T_ow0": f["T_ow0"][i],
                             "render": {"depth_fg": []}}
    cfg = W.c2_opt_cfg(max_iter=200)
    if kind == "ref":
        from oracle import ref_shim
        ns = ref_shim.import_reference()
        dec = ref_shim.build_reference_decoder(ns, p)
        opt = ns.optimizer.Optimizer({"device": "cpu", "opt": cfg, "vis": {"vis_pause_s": 0, "log_on": False, "vis_on": False}},
                                     dec, None, None)
        run = lambda z, T, pts: opt.shape_opt_deepsdf(z, T, pts, None)
    else:
        from oracle import hm_oracle as O
        od = O.fold_decoder(p)
        run = lambda z, T, pts: O.shape_opt_deepsdf(od, cfg, z, T, pts)
    _S[kind] = (MF.perturb, load, inp, run)
    return _S[kind]


def _run(task):
    kind, i, pert = task
    out = os.path.join(SCRATCH, f"{kind}_{i:03d}_{pert}.npz")
    if os.path.exists(out):
        return out
    import torch
    perturb, load_instance, inp, run = _setup(kind)
    d = perturb(load_instance(inp, i), pert)
    t = torch.from_numpy
    z, T, n = run(t(d["latent0"]), t(d["T_ow0"]), t(d["points_w"]))
    np.savez(out + ".tmp.npz", latent=z.detach().numpy(), iter_count=n)
    os.replace(out + ".tmp.npz", out)
    return out


def main():
    import multiprocessing as mp
    os.makedirs(SCRATCH, exist_ok=True)
    if NPTS != 1024 and not os.path.exists(os.path.join(HERE, INPUTS)):
        from hortimapping_amd import synthetic as S, workloads as W
        ds = W.make_c2_instances(S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3)), None, list(range(16)), kind="sdf")
        assert all(d["points_w"].shape == (NPTS, 3) for d in ds)
        np.savez_compressed(os.path.join(HERE, INPUTS), points_w=np.stack([d["points_w"] for d in ds]),
                            latent0=np.stack([d["latent0"] for d in ds]), T_ow0=np.stack([d["T_ow0"] for d in ds]))
    tasks = [("ref", i, p) for i in range(8) for p in REF_PERTS] + [("orc", i, p) for i in range(16) for p in ORC_PERTS]
    t0 = time.time()
    with mp.get_context("spawn").Pool(int(os.environ.get("WC_WORKERS", "7"))) as pool:
        for k, _ in enumerate(pool.imap_unordered(_run, tasks)):
            print(f"{k + 1}/{len(tasks)} runs, {time.time() - t0:.0f} s", flush=True)
    def gather(kind, n, perts):
        lat = np.zeros((len(perts), n, L), np.float32)
        for pi, p in enumerate(perts):
            for i in range(n):
                r = np.load(os.path.join(SCRATCH, f"{kind}_{i:03d}_{p}.npz"))
                assert int(r["iter_count"]) == 200
                lat[pi, i] = r["latent"]
        return lat
    np.savez_compressed(os.path.join(HERE, OUT), ref_latent=gather("ref", 8, REF_PERTS),
                        orc_latent=gather("orc", 16, ORC_PERTS), ref_perts=np.array(REF_PERTS), orc_perts=np.array(ORC_PERTS),
                        n_iter=200)
    print("written", OUT, flush=True)


if __name__ == "__main__":
    main()
