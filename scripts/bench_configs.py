"""Throughput of the shipped configurations at their real render-block sizes (L = 32 decoders, early exits on), on
synthetic fruits: the numbers a user of the reference would compare with its "0.6 s per fruit" (README.md:23).
Not the headline bench (that is bench.py / BASELINE.json configs[1]); printed as a small table for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CASES = [  # name, yaml, pose_known, instance shape
    ("wild_pepper (Sim3, 10 frames x 400 rays x 30, max_iter 50)", "wild_pepper.yaml", False,
     dict(n_pts=2000, n_frames=10, n_fg=200, n_bg=200)),
    ("challenge_pepper (pose known, 5 frames x 300 rays x 20, max_iter 20)", "shape_completion_challenge_pepper.yaml", True,
     dict(n_pts=2000, n_frames=5, n_fg=200, n_bg=100)),
    ("lab_berry (Sim3, 8 frames x 600 rays x 15)", "lab_berry.yaml", False,
     dict(n_pts=2000, n_frames=8, n_fg=400, n_bg=200, r_max=0.04)),
]
for name, y, known, shape in CASES:
    opt = yaml.safe_load(open(os.path.join(ROOT, "configs", y)))["opt"]
    r0 = 0.02 if "berry" in y else 0.04
    p = S.make_synthetic_decoder(32, seed=1, r0=r0, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(p)
    Ws, bs = S.fold_weight_norm(p)
    fac = W.gpu_sdf_factory(dec)
    protos = [S.make_instance(Ws, bs, 32, i, sdf_fn_factory=fac, **shape) for i in range(8)]
    insts = [W.to_instance(protos[i % 8], pose_known=known) for i in range(B)]
    for prec in ("f16x3", "f32"):
        dec.set_precision(prec)
        HO.optimize_batch(dec, opt, insts[:8])
        torch.cuda.synchronize()
        t = time.time()
        res = HO.optimize_batch(dec, opt, insts)
        torch.cuda.synchronize()
        dt = time.time() - t
        its = np.array([r.iter_count for r in res])
        print(f"{name} | {prec} | {B} fruits in {dt*1e3:.0f} ms = {dt/B*1e3:.2f} ms/fruit ({B/dt:.0f} fruits/s), "
              f"iterations mean {its.mean():.1f} (min {its.min()}, max {its.max()})", flush=True)
