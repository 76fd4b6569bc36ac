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
