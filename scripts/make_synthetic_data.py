#!/usr/bin/env python3
"""Write the synthetic stand-ins for the two datasets the reference downloads (scripts/download_*.sh):
data/synthetic_bup20 (BUP20 layout, 3 peppers) and data/synthetic_challenge/val (challenge layout), plus the two YAML
files that point the entry-point scripts at them.  Needs the GPU (scene rendering uses hm_decode_batch)."""
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hortimapping_amd import datasets as DS, synthetic as S, workloads as W      # noqa: E402
from hortimapping_amd.decoder import DecoderWeights                              # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "data")
params = S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(params)
fac = W.gpu_sdf_factory(dec)
DS.write_synthetic_bup20(os.path.join(out, "synthetic_bup20"), params, fac, n_fruits=3, n_frames=4)
DS.write_synthetic_challenge(os.path.join(out, "synthetic_challenge"), "val", params, fac, n_fruits=3, n_frames=5)
for src, dst, upd in (("wild_pepper.yaml", "synthetic_wild_pepper.yaml",
                       {"data_dir": os.path.join(out, "synthetic_bup20"),
                        "cam_info_path": os.path.join(out, "synthetic_bup20", "cam_info.yaml")}),
                      ("shape_completion_challenge_pepper.yaml", "synthetic_challenge_pepper.yaml",
                       {"data_dir": os.path.join(out, "synthetic_challenge"), "split": "val"})):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", src)))
    cfg.update(upd)
    cfg["deepsdf_dir"] = "synthetic:latent=32,seed=1,r0=0.04"
    cfg["vis"]["vis_on"] = False
    cfg["vis"]["log_on"] = False
    yaml.safe_dump(cfg, open(os.path.join(out, dst), "w"))
    print("wrote", os.path.join(out, dst))
