"""Experiment: the c2_joint batch (64 peppers, L = 256, 200 forced iterations) as ONE batch on one stream vs. split into
G groups, each on its own HIP stream (own workspace), enqueued from G host threads: the tail of one group's iteration
(render Jacobian launch, normal equations, solve: few CUs busy) can then run beside another group's main launch."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO
from hortimapping_amd.decoder import DecoderWeights
L, B, ITERS = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 200
params = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(params).set_precision("f16x3")
cfg = W.c2_opt_cfg(max_iter=ITERS, n_sample_on_ray=16, n_frame=1)
hcfg = HO.opt_cfg_from_dict(cfg)
dicts = W.make_c2_instances(params, dec, list(range(64)), kind="joint")
insts = [W.to_instance(dicts[i % 64]) for i in range(B)]

def build(groups, prio):
    out = []
    for g in range(groups):
        sub = insts[g * B // groups:(g + 1) * B // groups]
        if not sub: continue
        pb = HO.PackedBatch(sub, L, 1, "cuda")
        ws = HO.Workspace(dec, pb.B, pb.points_stride, pb.F, pb.R, hcfg.n_sample_on_ray)
        out.append(dict(pb=pb, ws=ws, init=(pb.latent.clone(), pb.T_ow.clone()), stream=torch.cuda.Stream(priority=prio)))
    return out

def run(gs, threaded):
    for g in gs:
        g["pb"].latent.copy_(g["init"][0]); g["pb"].T_ow.copy_(g["init"][1])
    torch.cuda.synchronize()
    t = time.perf_counter()
    def work(g):
        with torch.cuda.stream(g["stream"]):
            HO.run_packed(g["ws"], hcfg, g["pb"], 0)
    if threaded and len(gs) > 1:
        th = [threading.Thread(target=work, args=(g,)) for g in gs]
        [x.start() for x in th]; [x.join() for x in th]
    else:
        for g in gs: work(g)
    torch.cuda.synchronize()
    return time.perf_counter() - t

ref = None
for groups, threaded in ((1, False), (2, True), (3, True), (4, True), (6, True), (8, True), (16, True), (1, False), (4, True)):
    gs = build(groups, 0)
    run(gs, threaded)
    ts = [run(gs, threaded) for _ in range(3)]
    lat = torch.cat([g["pb"].latent for g in gs]).cpu()
    if ref is None: ref = lat
    print(f"{groups} group(s), {'threads' if threaded else 'one thread'}: {min(ts)*1e3:.1f} ms per optimisation of {B} -> {B/min(ts):.1f} instances/s; bits equal to one batch: {bool(torch.equal(lat, ref))}", flush=True)
    for g in gs: g["ws"].release()
