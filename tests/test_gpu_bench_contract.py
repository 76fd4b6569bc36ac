"""bench.py prints ONE JSON line with the fields the driver parses (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline).  A small
job (4 fruits, L = 32, 5 iterations) keeps this test fast; the numbers themselves are not asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--latent", "32", "--iters", "5"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert r["algorithmic_bytes_per_launch"] > 0 and "traffic_over_algorithmic" in r
    st = r["step"]                                   # whole-iteration algorithmic flop from the device-side counters
    cn = st["counts_per_step"]
    assert cn["instance_iterations"] == 4 * 5 and cn["N_J_sdf_term"] == 4 * 5 * 1024
    assert 0 < cn["N_F_ray_samples"] <= 4 * 5 * 1024 and 0 < cn["N_J_render"] <= cn["N_F_ray_samples"] and 0 < cn["V_rays"] <= 4 * 5 * 64
    E = 32 + 7
    A = ((cn["N_J_sdf_term"] + cn["N_J_render"]) * 7342080 + cn["N_F_ray_samples"] * 3671040
         + 2 * (cn["N_J_sdf_term"] + 2 * cn["V_rays"]) * E * E + cn["instance_iterations"] * (2.0 / 3.0) * E ** 3)
    assert abs(st["algorithmic_flop_per_step"] - A) <= 1e-9 * A and st["achieved"] > 0 and 0 < st["frac"] < 1
    assert d["rccl_ranks"] == 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] and c["sample"]
    assert abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 0.02 * d["value"]     # value = units / time


def test_bench_default_line_carries_the_secondary_objects():
    """The default `python bench.py` line (BASELINE.json configs[1], 64 peppers x 200 iterations; one timed step here)
    carries what SURVEY.md 8d asks beside the headline: the C2-sdf run, the whole-step roofline from device-side
    counters, one rank's share of configs[3], and the labelled other arithmetics."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dtype"] == "f16x3" and d["config"]["instances_total"] == 64 and d["config"]["iterations"] == 200
    for k in ("exact_f32", "mixed_f16x3f_f16b", "plain_f16", "trained_decoder", "batch_256", "c2_sdf", "configs3_rank_share"):
        assert k in d and d[k]["value"] > 0, k
    st = d["roofline"]["step"]
    cn = st["counts_per_step"]
    assert cn["instance_iterations"] == 64 * 200 and cn["N_J_sdf_term"] == 64 * 200 * 1024
    assert st["algorithmic_flop_per_step"] > d["roofline"]["algorithmic_flop_per_launch"] * 200
    sd = d["c2_sdf"]["roofline"]["step"]["counts_per_step"]
    assert sd["N_J_sdf_term"] == 64 * 200 * 2048 and sd["N_F_ray_samples"] == 0 and sd["V_rays"] == 0
    assert d["configs3_rank_share"]["instances"] == 512 and d["configs3_rank_share"]["roofline_step"]["counts_per_step"]["instance_iterations"] == 512 * 200
    assert all(d[k]["steps"] >= 3 for k in ("c2_sdf", "exact_f32", "mixed_f16x3f_f16b", "plain_f16", "trained_decoder", "batch_256", "configs3_rank_share"))
    # the configurations users run (BASELINE.json configs[0], [2], [4]) at their real render-block sizes, >= 3 timed steps each
    for k, n_groups in (("configs0_wild_pepper", 1), ("configs2_challenge_pepper", 1), ("configs4_lab_pepper_berry", 2)):
        o = d[k]
        assert o["value"] > 0 and o["ms_per_fruit"] > 0 and o["steps"] >= 3 and o["fruits"] == 64 and o["dtype"] == "f16x3", k
        assert len(o["groups"]) == n_groups and 0 < o["roofline_step"]["frac"] < 1 and o["roofline_step"]["algorithmic_flop_per_step"] > 0
        for g in o["groups"]:
            c = g["counts_per_step"]
            assert c["instance_iterations"] >= 3 * g["fruits"] and c["N_F_ray_samples"] > 0 and c["V_rays"] > 0
            assert 3 <= g["iterations"]["min"] <= g["iterations"]["max"] <= g["max_iter"]
    assert d["configs4_lab_pepper_berry"]["plain_f16"]["dtype"] == "f16" and d["configs4_lab_pepper_berry"]["plain_f16"]["value"] > 0


def test_bench_strong_scaling_configs3_rank_share():
    """BASELINE.json configs[3] as one rank of eight sees it: 512 instances at the benchmark size (L = 256, 1024 surface
    points + 64 rays x 16 samples) through ONE workspace in two chunks of 256, `bench.py --total` (the full job is
    --total 4096 over 8 GPUs; the iteration count is cut to keep the test short)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--total", "512", "--batch", "256", "--iters", "6", "--no-exact", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0
    c = d["config"]
    assert c["instances_total"] == 512 and c["instances_per_gpu"] == 512 and c["chunk"] == 256 and c["iterations"] == 6
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    assert d["roofline"]["launches"] == 2 * 6                          # two chunks x six iterations were profiled


def test_bench_two_ranks_real_job_on_one_gpu():
    """bench.py's N > 1 branch with the REAL optimisation: two ranks launched like the driver does (one process per
    rank, RANK / WORLD_SIZE / MASTER_* in the environment), both on this box's single GPU with the collectives over gloo
    (`--share-gpu`; RCCL cannot place two ranks on one device).  Checks what the 8-GPU run relies on: the shard plan,
    per-rank instance generation, the record all-gather in instance order, the MAX all-reduce of the time, one JSON line
    from rank 0 only -- and that the gathered results equal a single-rank run of the same 2 x 4 instances."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    common = ["--steps", "1", "--warmup", "0", "--batch", "4", "--latent", "32", "--iters", "5", "--no-exact",
              "--no-cpu-baseline"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu",
                                       "--dump-records", os.path.join(ROOT, "gpurun_out", "tmp_records_2rank.pt")] + common,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["instances_total"] == 8 and "test_mode" in d
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-records",
                          os.path.join(ROOT, "gpurun_out", "tmp_records_1rank.pt"), "--steps", "1", "--warmup", "0", "--batch",
                          "8", "--latent", "32", "--iters", "5", "--no-exact", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-1500:]
    import torch
    a = torch.load(os.path.join(ROOT, "gpurun_out", "tmp_records_2rank.pt"))
    b = torch.load(os.path.join(ROOT, "gpurun_out", "tmp_records_1rank.pt"))
    assert a.shape == b.shape == (8, 32 + 18) and torch.equal(a, b)          # same instances, same order, same bits


def test_bench_started_bare_spawns_ranks_that_run_the_real_job():
    """`python bench.py --gpus 2` with no launcher and no rank environment (how the driver starts the N = 1 bench, with a
    larger N): bench.py re-executes itself under torch.distributed.run.  Here the two ranks share this box's one GPU
    (`--share-gpu`, collectives over gloo; on a multi-GPU node the same path runs RCCL, tests/test_gpu_rccl.py); the
    gathered records must equal the one-rank run bit for bit and exactly one JSON line may come out."""
    import torch
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID",
                                                              "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "1", "--warmup", "0", "--latent", "32", "--iters", "5", "--no-exact", "--no-cpu-baseline"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--batch", "4",
                          "--dump-records", os.path.join(out_dir, "tmp_records_bare2.pt")] + common, env=env,
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-1500:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo" and "test_mode" in d
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--dump-records",
                          os.path.join(out_dir, "tmp_records_bare1.pt")] + common, env=env, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-1500:]
    a = torch.load(os.path.join(out_dir, "tmp_records_bare2.pt"))
    b = torch.load(os.path.join(out_dir, "tmp_records_bare1.pt"))
    assert a.shape == b.shape == (8, 32 + 18) and torch.equal(a, b)
