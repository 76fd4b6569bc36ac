"""world_size-2 CPU test (gloo) of the N > 1 path: contiguous instance sharding + the single all-gather of result
records, with uneven shards.  The per-rank 'optimiser' is a deterministic stand-in (the real one needs a GPU); what is
tested is exactly the code bench.py and optimize_sharded run around it."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, L, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hortimapping_amd import distributed as D
    r, lr, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    def run_local(lo, hi):
        ids = torch.arange(lo, hi, dtype=torch.float32)
        lat = ids[:, None] * 1000 + torch.arange(L, dtype=torch.float32)[None]
        T = ids[:, None].repeat(1, 16) + 0.5
        return lat, T, (ids.int() % 7), (ids.int() % 3) * 8
    lat, T, it, st = D.optimize_sharded(run_local, n_total, L, torch.device("cpu"))
    q.put((rank, lat.clone(), T.clone(), it.clone(), st.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_preserves_instance_order():
    world, n_total, L = 2, 7, 32            # 7 instances over 2 ranks: shards of 4 and 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = torch.arange(n_total, dtype=torch.float32)
    for rank, lat, T, it, st in outs:
        assert lat.shape == (n_total, L) and T.shape == (n_total, 4, 4)
        assert torch.equal(lat[:, 0], ids * 1000) and torch.equal(lat[:, 5], ids * 1000 + 5)
        assert torch.equal(T[:, 0, 0], ids + 0.5)
        assert torch.equal(it, ids.int() % 7) and torch.equal(st, (ids.int() % 3) * 8)


def _run_bench_ranks(world, extra, port):
    """Launch bench.py's own main() once per rank (gloo, `--stub-cpu`: the GPU optimisation is replaced by a stand-in,
    everything around it -- argument check, shard plan, chunk loop, gather, MAX all-reduce, JSON -- is the real code)."""
    import json
    import subprocess
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu", "--gpus", str(world),
                                       "--latent", "32", "--steps", "2", "--warmup", "1"] + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly ONE JSON line, from rank 0
    assert all(not l.startswith("{") for so, _ in outs[1:] for l in so.splitlines())
    return json.loads(lines[0])


def test_bench_rank_logic_two_ranks_strong_and_weak_scaling():
    port = 29900 + (os.getpid() % 90)
    # strong scaling (configs[3] in miniature): 37 instances over 2 ranks (19 + 18) in chunks of 8
    j = _run_bench_ranks(2, ["--total", "37", "--batch", "8"], port)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["instances_total"] == 37 and j["config"]["instances_per_gpu"] == 19 and j["config"]["chunk"] == 8
    assert j["value"] > 0 and abs(j["value"] - 37 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-2 * j["value"]
    assert "stub" in j and "roofline" not in j               # the stand-in never passes for a measurement
    # weak scaling: 64 per rank
    j = _run_bench_ranks(2, [], port + 1)
    assert j["scaling"] == "weak" and j["config"]["instances_total"] == 128 and j["config"]["instances_per_gpu"] == 64


def test_bench_rank_logic_eight_ranks_configs3_plan_and_weak_scaling():
    """The driver's 8-GPU runs in dry form (8 gloo ranks, `--stub-cpu`): BASELINE.json configs[3] = `--gpus 8 --total 4096`
    (512 instances per rank in chunks [256, 256], records gathered in instance order -- asserted inside bench.py on the
    gathered matrix) and the weak-scaling default (64 per rank, 512 in total)."""
    port = 29700 + (os.getpid() % 90)
    j = _run_bench_ranks(8, ["--total", "4096"], port)
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["scaling"] == "strong"
    c = j["config"]
    assert c["instances_total"] == 4096 and c["instances_per_gpu"] == 512 and c["chunk"] == 256
    assert abs(j["value"] - 4096 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-2 * j["value"]
    j = _run_bench_ranks(8, [], port + 1)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["instances_total"] == 512 and j["config"]["instances_per_gpu"] == 64


def test_ipc_mode_is_set_for_every_launch_path():
    """RCCL across processes needs dmabuf IPC on these hosts: importing the distributed module sets the variable (unless the
    caller exported one), so a rank started by torch.distributed.run directly gets it like one started by bench.py."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    code = "import os, sys; sys.path.insert(0, %r); import hortimapping_amd.distributed; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip() == "0"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip() == "1"


def test_bench_refuses_a_gpu_count_that_does_not_match_the_launch():
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu", "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_bench_spawns_its_own_ranks_when_started_bare():
    """`python bench.py --gpus 2` with no launcher and no rank environment (the shape of the driver's N = 1 command with
    a larger N) re-executes itself under torch.distributed.run: two ranks, one JSON line, the process-group size in it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID",
                                                              "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu", "--gpus", "2", "--latent", "32",
                        "--steps", "2", "--warmup", "1", "--total", "37", "--batch", "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["collective_backend"] == "gloo" and j["scaling"] == "strong"
    assert j["config"]["instances_total"] == 37 and j["config"]["instances_per_gpu"] == 19


def test_bench_shard_plan():
    sys.path.insert(0, ROOT)
    import bench
    ids, chunks, n = bench.plan_shard(3, 8, 64, 0, 256)                  # weak: rank 3 of 8
    assert ids == list(range(192, 256)) and chunks == [(0, 64)] and n == 512
    seen = []
    for r in range(8):                                                   # configs[3]: 4096 over 8 GPUs, chunks of 256
        ids, chunks, n = bench.plan_shard(r, 8, 64, 4096, 256)
        assert n == 4096 and len(ids) == 512 and chunks == [(0, 256), (256, 512)]
        seen += ids
    assert seen == list(range(4096))
    ids, chunks, n = bench.plan_shard(1, 2, 64, 37, 8)                   # uneven: 19 + 18
    assert ids == list(range(19, 37)) and chunks == [(0, 8), (8, 16), (16, 18)]
