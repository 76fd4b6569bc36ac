"""Instance sharding over the GPUs of one node + the single result gather (RCCL over xGMI on GPUs, gloo on CPU).

Fruit instances are independent (each `shape_pose_joint_opt` call of the reference touches only its own latent, pose
and observations, wild_completion/optimizer.py:28-302), so the N-GPU path is: contiguous block partition of the
instance list (keeps "identical instance indexing"), zero communication during the optimisation, and ONE all-gather of
a fixed-size record per instance at the end:  [latent (L) | T_ow (16) | iter_count | status]  fp32.
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

# RCCL between PROCESSES maps peer buffers through HIP IPC handles.  The hosts this code runs on only support dmabuf IPC
# (their environment exports HSA_ENABLE_IPC_MODE_LEGACY=0; with the legacy mode RCCL's `hipIpcGetMemHandle` fails with
# "invalid argument" -- the platform notes of this build state it, no multi-GPU box was available to re-measure it).  The
# variable is read when the HSA runtime starts, so it is set here, at import, before any rank touches the GPU -- for every
# launch path alike (torch.distributed.run started by the driver, bench.py's own spawn, a user's script).  A value the
# caller exported wins.
def _set_ipc_mode():
    """setdefault + a warning when it comes too late: the HSA runtime reads the variable when it starts, so a process that
    has already initialised the GPU (torch.cuda touched before this import) keeps whatever mode it started with."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
        return
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        import warnings
        warnings.warn("hortimapping_amd.distributed was imported after the GPU runtime started: HSA_ENABLE_IPC_MODE_LEGACY=0 "
                      "could not take effect in this process; multi-process RCCL may fail with 'hipIpcGetMemHandle: invalid "
                      "argument'.  Export HSA_ENABLE_IPC_MODE_LEGACY=0 before starting Python (or import hortimapping_amd "
                      "before touching torch.cuda).", RuntimeWarning, stacklevel=3)


_set_ipc_mode()


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Instance i belongs to rank i // ceil(n / world): contiguous blocks, order preserving."""
    per = math.ceil(n / world) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def pack_records(latent: torch.Tensor, T_ow: torch.Tensor, iter_count: torch.Tensor, status: torch.Tensor) -> torch.Tensor:
    """(n, L+18) fp32 record matrix; ints are exactly representable (< 2^24)."""
    n = latent.shape[0]
    return torch.cat([latent.reshape(n, -1).float(), T_ow.reshape(n, 16).float(),
                      iter_count.reshape(n, 1).float(), status.reshape(n, 1).float()], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, L: int):
    return rec[:, :L], rec[:, L:L + 16].reshape(-1, 4, 4), rec[:, L + 16].round().int(), rec[:, L + 17].round().int()


def gather_records(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather the per-rank record blocks into the global (n_total, W) matrix, in instance order.
    One collective: ranks pad their block to ceil(n_total / world) rows (all_gather_into_tensor needs equal sizes)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    per = math.ceil(n_total / world)
    W = local.shape[1]
    buf = torch.zeros(per, W, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = torch.empty(world * per, W, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return out[:n_total]


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def optimize_sharded(run_local, n_total: int, L: int, device) -> Tuple[torch.Tensor, ...]:
    """`run_local(lo, hi)` optimises instances [lo, hi) and returns (latent, T_ow, iter_count, status) tensors on
    `device`; every rank gets the results of ALL instances back, in instance order."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, rank, world)
    lat, T, it, st = run_local(lo, hi)
    rec = pack_records(lat.to(device), T.to(device), it.to(device), st.to(device))
    allrec = gather_records(rec, n_total)
    return unpack_records(allrec, L)
