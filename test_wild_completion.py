#!/usr/bin/env python3
"""Entry point #1 of the reference, kept as the drop-in surface:  python test_wild_completion.py -c <yaml>

Same option, same YAML schema, same input layout and same output files as the reference's
`test_wild_completion.py:23-264` (BUP20 "wild" sequences: per-submap completion + pose files), but ALL fruit instances
of the sequence are optimised in one batched call on the MI355X (`Optimizer.optimize_batch`) instead of one after the
other.  Skipping rules are the reference's, applied before batching, so results stay keyed by submap file name:
Background submap, ids below `begin_submap`, submaps without a matched frame (`get_render_data`), invalid pose
initialisation, and the post-hoc outlier rejection on scale / pitch / roll.

Output contract of the completed meshes (`submaps_complete/<name>.ply`): the zero level set of the decoded SDF grid the
reference hands to scikit-image's marching cubes (`wild_completion/utils.py:565-588`), extracted here by marching cubes
on the GPU (`hm_extract_surface_mc`): the vertices are exactly the grid-edge crossings any marching-cubes variant
produces; the triangulation of ambiguous cells comes from a generated table and may differ from scikit-image's
Lewiner tables (`tests/test_gpu_mesher.py`, `oracle/level_set.py`).

`deepsdf_dir` may be a DeepSDF experiment directory (specs.json + ModelParameters + LatentCodes, as in the reference)
or `synthetic:latent=<L>,seed=<s>[,r0=<r>]` for the analytic decoder (the reference tree ships no weights)."""
import os

import click
import numpy as np
import torch
import yaml
from numpy.linalg import det, inv

from hortimapping_amd import data_prep as DP, datasets as DS, synthetic as S
from hortimapping_amd.decoder import DecoderWeights, config_decoder, load_latent_vectors
from hortimapping_amd.mesher import MeshExtractor, read_ply, write_ply
from hortimapping_amd.optimizer import Instance, Optimizer, STATUS_INVALID
from hortimapping_amd.utils import StageTimer


def load_decoder(cfg):
    d = cfg["deepsdf_dir"]
    if isinstance(d, str) and d.startswith("synthetic:"):
        kv = dict(x.split("=") for x in d[len("synthetic:"):].split(","))
        L = int(kv.get("latent", 32))
        params = S.make_synthetic_decoder(L, seed=int(kv.get("seed", 1)), r0=float(kv.get("r0", 0.04)),
                                          aniso=(1.0, 0.75, 1.3))
        return DecoderWeights.from_params(params), torch.zeros(L), params
    decoder = config_decoder(d, "latest")                               # test_wild_completion.py:44-45
    init_latent = torch.mean(load_latent_vectors(d, "latest"), 0)       # :46-47
    return decoder, init_latent, None


@click.command()
@click.option("--config", "-c", type=str, help="path to the config file (.yaml)",
              default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs/wild_pepper.yaml"))
@click.option("--dump-jobs", type=str, default="", hidden=True,
              help="(tests) torch.save the prepared per-instance inputs and the raw optimiser results here")
def main(config, dump_jobs):
    np.random.seed(42)                                                  # set_random_seed(42), utils.py:638-641
    torch.manual_seed(42)
    timer = StageTimer()                                                # off unless HM_STAGE_TIMES names a file
    cfg = yaml.safe_load(open(config))
    dtype = torch.float32
    decoder, init_latent, _ = load_decoder(cfg)
    code_len = init_latent.shape[0]
    print("DeepSDF model loaded")

    data_base = cfg["data_dir"]
    submap_folder = os.path.join(data_base, "submaps")
    complete_submap_folder, clean_submap_folder, pose_folder = (submap_folder + s for s in ("_complete", "_clean", "_pose"))
    for d in (complete_submap_folder, clean_submap_folder, pose_folder):
        os.makedirs(d, exist_ok=True)

    object_radius_max_m = float(cfg["vis"]["object_radius_max_m"])
    voxels_dim = int(2 * object_radius_max_m * 1e3 / float(cfg["vis"]["mc_res_mm"]))
    K, _, img_size = DS.load_cam_info(cfg["cam_info_path"])
    invK = inv(K)
    frames = DS.load_bup20_frames(cfg)
    print("Loaded %d frames, image size %s" % (len(frames["id"]), img_size))

    mesh_extractor = MeshExtractor(decoder, code_len=code_len, voxels_dim=voxels_dim, cube_radius=object_radius_max_m)
    if not os.environ.get("HM_PRECISION"):
        # fp32-class arithmetic at three times the speed of exact fp32; an instance whose activations leave the fp16 range
        # is rerun in exact fp32 by the Optimizer itself (hortimapping_amd/optimizer.py: retry_f32)
        decoder.set_precision("f16x3")
    opt = Optimizer(cfg, decoder, mesh_extractor, None)
    timer.lap("load decoder + frames")

    # instance loop of the reference (:133), split in two so that the image scans of get_render_data run on the GPU for
    # all instances at once; the np.random.choice draws inside keep the reference's order (submap by submap)
    entries, bg_at = [], {}
    for submap_name in sorted(os.listdir(submap_folder)):
        submap_cat = (submap_name.split("_")[1]).split(".")[0]
        submap_id = int(submap_name.split("_")[0])
        if submap_id > 1 and submap_id < cfg["begin_submap"]:
            continue
        cur_mesh = read_ply(os.path.join(submap_folder, submap_name))
        if submap_cat == "Background":                                   # :148-151 -- seen by the submaps AFTER it
            bg_at[len(entries)] = DP.voxel_down_sample(cur_mesh.sample_points_uniformly(500000, seed=42), 0.005)
            continue
        entries.append((submap_name, submap_id, cur_mesh))
    timer.lap("read submap meshes (+ background cloud)")
    dev_frames = DP.DeviceFrames(frames["id"], frames["depth"])
    render_all = DP.get_render_data_device([e[1] for e in entries], dev_frames, frames["pose"], img_size, invK, cfg)

    # clean_pcd (DBSCAN) of all matched submaps in one launch, the background crops of get_pose_init per background cloud
    matched = [k for k, rd in enumerate(render_all) if rd["count"] > 0]
    for k, rd in enumerate(render_all):
        if rd["count"] == 0:
            print("Submap %d: no valid match, skip" % entries[k][1])
    rc = cfg["opt"]["recon"]
    samples = [entries[k][2].sample_points_uniformly(rc["n_pts"], seed=42) for k in matched]     # clean_mesh, utils.py:389-405
    cleaned = dict(zip(matched, DP.clean_pcd_device(samples, rc["cluster_dist_m"]))) if matched else {}
    boxes = {k: DP.pose_init_box(cleaned[k]) for k in matched}
    crops, bg_cloud, pending = {}, None, []

    def flush():
        if pending and bg_cloud is not None and len(bg_cloud):
            for k, c in zip(pending, bg_cloud.crop_boxes([boxes[k][3] for k in pending], [boxes[k][4] for k in pending])):
                crops[k] = c
        pending.clear()
    for k in range(len(entries) + 1):
        if k in bg_at:                                                   # a Background submap precedes entry k
            flush()
            bg_cloud = DP.DeviceCloud(bg_at[k])
        if k in boxes and boxes[k][2]:
            pending.append(k)
    flush()

    jobs = []
    for k in matched:
        submap_name, submap_id, _ = entries[k]
        pts = cleaned[k]
        center, bbx_size, valid, _, _ = boxes[k]
        if not valid:
            print("Submap %d: invalid pose initialisation, skip" % submap_id)
            continue
        rot_y = DP.pose_init_rotation(center, crops.get(k))
        T_wo = DP.init_T_wo(center, rot_y, bbx_size, cfg["opt"], object_radius_max_m)
        inst = Instance(init_latent.clone(), torch.tensor(inv(T_wo), dtype=dtype), torch.tensor(pts, dtype=dtype),
                        render_all[k], object_radius_max_m, False)
        jobs.append((submap_name, submap_id, pts, inst))
    timer.lap("device data prep (render data, DBSCAN, pose init)")
    print("Optimising %d fruit instances in one batch" % len(jobs))
    results = opt.optimize_batch([j[3] for j in jobs]) if jobs else []
    timer.lap("optimise (pack + upload + LM loop + download)")
    if dump_jobs:                     # tests/test_gpu_cli.py feeds exactly these inputs to the CPU oracle
        DS.dump_jobs(dump_jobs, [(j[0], j[3]) for j in jobs], results, cfg["opt"], opt.decoder.precision)

    kept = 0
    checks = []
    for (submap_name, submap_id, pts, _), res in zip(jobs, results):
        if res.status & STATUS_INVALID:
            print("Submap %d: not valid (no depth residuals)" % submap_id)
        checks.append(DP.final_pose_check(res.T_ow.numpy().astype(np.float64), cfg["opt"]["outlier"]))
    # the completed meshes of ALL kept fruits: one batched grid decode + one marching-cubes launch (the reference meshes
    # fruit by fruit inside its loop, :249; same meshes either way)
    keep_idx = [k for k, c in enumerate(checks) if c[3]]
    meshes = dict(zip(keep_idx, mesh_extractor.extract_meshes(torch.stack([results[k].latent for k in keep_idx])))) if keep_idx else {}
    timer.lap("grid decode + marching cubes (all kept fruits, batched)")
    for k, ((submap_name, submap_id, pts, _), res) in enumerate(zip(jobs, results)):
        T_wo, final_scale, (yaw, pitch, roll), keep = checks[k]
        if not keep:                                                              # :238-246
            print("Submap %d: final scale %f / pitch %f / roll %f is an outlier, not valid"
                  % (submap_id, final_scale, pitch, roll))
            continue
        mesh = meshes[k].transform(T_wo)                                          # complete_mesh, mesher.py:26-32
        write_ply(mesh, os.path.join(complete_submap_folder, submap_name))       # :249-252
        DS.write_points_ply(pts, os.path.join(clean_submap_folder, submap_name))  # :254-256
        np.save(os.path.join(pose_folder, submap_name.replace("ply", "npy")), T_wo)   # :258-260
        kept += 1
        print("Submap %d: %d iterations, scale %.3f -> %s" % (submap_id, res.iter_count, final_scale,
                                                              os.path.join(complete_submap_folder, submap_name)))
    timer.lap("weld check + write .ply / .npy")
    timer.write(script="test_wild_completion.py", fruits=len(jobs), kept=kept,
                mean_iterations=float(np.mean([r.iter_count for r in results])) if results else 0.0)
    print("Completed %d of %d submaps" % (kept, len(jobs)))


if __name__ == "__main__":
    main()
