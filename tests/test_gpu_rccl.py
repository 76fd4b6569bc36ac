"""The RCCL leg of the N-GPU path on the one GPU a test box has: a 1-rank `nccl` process group (RCCL on ROCm) runs the
very collective sequence bench.py and optimize_sharded use (barrier, all_gather_into_tensor of the record matrix,
MAX all-reduce of the step time) on device tensors.  The multi-rank logic itself (uneven shards, instance order) is
covered on CPU by test_distributed_gloo.py; multi-GPU runs are the driver's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from hortimapping_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
L, n = 32, 5
def run_local(lo, hi):
    ids = torch.arange(lo, hi, dtype=torch.float32, device="cuda")
    return (ids[:, None] * 10 + torch.arange(L, device="cuda")[None], ids[:, None].repeat(1, 16),
            ids.int() + 3, ids.int() * 8)
dist.barrier()
lat, T, it, st = D.optimize_sharded(run_local, n, L, torch.device("cuda"))
t = torch.tensor([1.25], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert lat.is_cuda and lat.shape == (n, L) and float(lat[4, 7]) == 47.0 and int(it[2]) == 5 and int(st[3]) == 24
assert float(t) == 1.25
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
""" % ROOT


@pytest.mark.gpu
def test_single_rank_rccl_collectives():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_two_rank_rccl_real_job_on_two_devices():
    """Auto-enables on a box with >= 2 GPUs (skipped on the 1-GPU test boxes): `python bench.py --gpus 2` started bare,
    so it spawns its two ranks itself (torch.distributed.run, one process per device, backend "nccl" = RCCL over
    xGMI), each running the REAL optimisation of its shard; the gathered records must equal a single-rank run of the same
    2 x 4 instances bit for bit (instances are independent: sharding may not change a single bit), and the JSON line must
    report a 2-rank RCCL group.  This validates RCCL with N > 1 before any scaling number is trusted."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL cannot place two ranks on one device)")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    common = ["--steps", "1", "--warmup", "0", "--latent", "32", "--iters", "5", "--no-exact", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID",
                                                              "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--dump-records",
                          os.path.join(out_dir, "tmp_records_rccl2.pt")] + common, env=env, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-1500:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "nccl" and "test_mode" not in d
    assert d["config"]["instances_total"] == 8
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--dump-records",
                          os.path.join(out_dir, "tmp_records_rccl1.pt")] + common, env=env, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-1500:]
    a = torch.load(os.path.join(out_dir, "tmp_records_rccl2.pt"))
    b = torch.load(os.path.join(out_dir, "tmp_records_rccl1.pt"))
    assert a.shape == b.shape == (8, 32 + 18) and torch.equal(a, b)
