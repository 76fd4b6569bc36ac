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
