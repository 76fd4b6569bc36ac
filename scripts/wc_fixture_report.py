"""GPU box: per instance of the well-conditioned fixture, the oracle's own perturbation noise and the GPU deviations (f32,
f16x3) as fractions of the outright 1e-4 tolerance.  Used once to prune the fixture (tests/golden/make_wc_records.py prune)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hortimapping_amd import metrics as MX, optimizer as HO, workloads as W
from hortimapping_amd.decoder import DecoderWeights
G = os.path.join(ROOT, "tests", "golden")
inp, rec = np.load(os.path.join(G, "wc_fullsize_inputs.npz")), np.load(os.path.join(G, "wc_fullsize_oracle.npz"))
params = W.wc_decoder_params(256)
sampler = DecoderWeights.from_params(params); sampler.set_precision("f32")
gt = MX.ground_truth_points_world(sampler, inp["z_true"], inp["T_wo_true"])
m = np.stack([MX.completion_metrics(sampler, rec["free_latent"][p], rec["free_T_ow"][p], gt, inp["T_wo_true"]) for p in range(5)])
n = m.shape[1]
scale = np.stack([m[0][:, 0], np.maximum(m[0][:, 1], 1e-3), np.maximum(m[0][:, 2], 0.1), np.ones(n)], axis=1)
tol = 1e-4 * scale
noise = (np.abs(m[1:] - m[0]).max(axis=0) / tol)
out = {"inst_ids": inp["inst_ids"].tolist(), "oracle_noise_frac": noise.max(axis=1).tolist()}
for prec in ("f32", "f16x3"):
    dec = DecoderWeights.from_params(params); dec.set_precision(prec)
    res = HO.optimize_batch(dec, W.wc_opt_cfg(200), [W.to_instance(d) for d in W.fixture_dicts(inp)])
    mg = MX.completion_metrics(sampler, torch.stack([r.latent for r in res]).numpy(), [r.T_ow.numpy() for r in res], gt, inp["T_wo_true"])
    out[prec] = (np.abs(mg - m[0]) / tol).max(axis=1).tolist()
for i in range(n):
    print(f"pos {i:2d} cand {out['inst_ids'][i]:3d}: oracle noise {out['oracle_noise_frac'][i]:.2f} (per metric {np.round(noise[i], 2).tolist()})  gpu f32 {out['f32'][i]:.2f}  f16x3 {out['f16x3'][i]:.2f}")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wc_fixture_report.json"), "w"))
