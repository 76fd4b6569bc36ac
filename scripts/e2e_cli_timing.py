#!/usr/bin/env python3
"""End-to-end throughput of the two entry points (VERDICT r04 missing #5 / next #6b), GPU box:

    python scripts/e2e_cli_timing.py [n_fruits=64] > gpurun_out/r05_e2e_cli.json

Writes a 64-fruit synthetic BUP20 sequence (10 frames of 720 x 1280, a wall of fruits; reference layout
`test_wild_completion.py:60-131`) and a 64-fruit shape-completion-challenge split (5 frames per fruit; layout of
`run_shape_completion_challenge.py:93-170`) under /tmp, then runs each script TWICE as a user would (`python <script> -c
<yaml>`, a fresh process: import, library load, decoder build, data read, device data preparation, ONE batched
optimisation, grid decode + marching cubes, PLY / pose / metric output) with HM_STAGE_TIMES on, and reports the second run:
fruits/s over the whole process, the wall-time split by stage, and the largest stage.  The reference quotes 0.6 s per
fruit for the wild pipeline on an unnamed CUDA GPU (README.md:23)."""
import json
import os
import subprocess
import sys
import time

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_data(out, n):
    from hortimapping_amd import datasets as DS, synthetic as S, workloads as W
    from hortimapping_amd.decoder import DecoderWeights
    params = S.make_synthetic_decoder(32, seed=1, r0=0.04, aniso=(1.0, 0.75, 1.3))
    dec = DecoderWeights.from_params(params)
    fac = W.gpu_sdf_factory(dec)
    t = time.time()
    DS.write_synthetic_bup20(os.path.join(out, "bup20"), params, fac, n_fruits=n, n_frames=10, img_size=(720, 1280))
    DS.write_synthetic_challenge(os.path.join(out, "challenge"), "val", params, fac, n_fruits=n, n_frames=5)
    cfgs = {}
    for src, key, upd in (("wild_pepper.yaml", "wild", {"data_dir": os.path.join(out, "bup20"),
                                                          "cam_info_path": os.path.join(out, "bup20", "cam_info.yaml")}),
                          ("shape_completion_challenge_pepper.yaml", "challenge",
                           {"data_dir": os.path.join(out, "challenge"), "split": "val"})):
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", src)))
        cfg.update(upd)
        cfg["deepsdf_dir"] = "synthetic:latent=32,seed=1,r0=0.04"
        cfg["vis"]["vis_on"] = False
        cfg["vis"]["log_on"] = False
        cfgs[key] = os.path.join(out, key + ".yaml")
        yaml.safe_dump(cfg, open(cfgs[key], "w"))
    return cfgs, time.time() - t


def run(script, cfg_path, tag, out):
    rec = None
    for rep in range(2):
        stage = os.path.join(out, f"{tag}_stages_{rep}.json")
        env = dict(os.environ, HM_STAGE_TIMES=stage)
        t = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), "-c", cfg_path], cwd=ROOT, env=env,
                           capture_output=True, text=True)
        wall = time.perf_counter() - t
        if r.returncode != 0:
            raise SystemExit(f"{script} failed:\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}")
        rec = json.load(open(stage))
        rec["process_wall_s"] = round(wall, 3)
        rec["stdout_tail"] = r.stdout.strip().splitlines()[-6:]
    st = rec["stages_s"]
    rec["in_script_s"] = rec.pop("total_s")
    rec["python_start_and_imports_s"] = round(rec["process_wall_s"] - rec["in_script_s"], 3)
    rec["largest_stage"] = max(st, key=st.get)
    n = max(1, rec["fruits"])
    rec["fruits_per_s_whole_process"] = round(n / rec["process_wall_s"], 2)
    rec["fruits_per_s_in_script"] = round(n / rec["in_script_s"], 2)
    rec["ms_per_fruit_whole_process"] = round(1e3 * rec["process_wall_s"] / n, 2)
    return rec


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out = "/tmp/hm_e2e"
    os.makedirs(out, exist_ok=True)
    cfgs, t_data = make_data(out, n)
    res = {"what": "second of two fresh-process runs of each entry point on synthetic data in the reference's layouts; "
                   "f16x3 arithmetic with the exact-f32 retry (the scripts' default)",
           "n_fruits_written": n, "data_generation_s": round(t_data, 1),
           "reference_published": "0.6 s per fruit, wild pipeline, unnamed CUDA GPU (README.md:23)"}
    res["test_wild_completion"] = run("test_wild_completion.py", cfgs["wild"], "wild", out)
    res["run_shape_completion_challenge"] = run("run_shape_completion_challenge.py", cfgs["challenge"], "challenge", out)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
