"""Multi-GPU plumbing: one process per GPU, replicas sharded by rank, ONE collective (statistics all-reduce).

The reference has no distributed backend (its fan-out is multiprocessing.Pool over whole simulations,
examples/JOCN_Benchmark_2024/graph_load.py:361-363); replicas are independent, so nothing is exchanged on the data
path. `backend="nccl"` is RCCL on ROCm; the same code runs on gloo for the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np


def init_process_group(backend: str = "nccl", local_rank: int = 0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def shard_bounds(global_batch: int, rank: int, world: int):
    """(first global replica, replica count) of `rank` when `global_batch` independent replicas are dealt to `world`
    ranks in contiguous slices (the first `global_batch % world` ranks take one more).  With
    `env.seed(seed, replica_base=first)` replica r of the rank draws stream (seed, first + r): a sharded run simulates
    exactly the replicas of the unsharded one, whatever the world size."""
    global_batch, rank, world = int(global_batch), int(rank), int(world)
    if not 0 <= rank < world or global_batch < world:
        raise ValueError("need 0 <= rank < world <= global_batch")
    q, rem = divmod(global_batch, world)
    first = rank * q + min(rank, rem)
    return first, q + (1 if rank < rem else 0)


def reduce_run_statistics(delta: np.ndarray, dt: float, kernel_ms: float, dist=None, device: str = "cuda"):
    """SUM of the per-rank counter deltas, MAX of the per-rank wall time and kernel time."""
    if dist is None:
        return np.asarray(delta, np.float64), float(dt), float(kernel_ms)
    import torch
    t = torch.tensor(np.asarray(delta, np.float64), device=device)
    dist.all_reduce(t)
    tt = torch.tensor([dt, kernel_ms], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return t.cpu().numpy(), float(tt[0]), float(tt[1])


def gather_per_rank(values, dist=None, device: str = "cuda"):
    """every rank's `values` (a short list of floats) on every rank, as a list of lists indexed by rank: lets the bench line
    show what the process group really held (its size, each rank's own rate) instead of what the launcher's environment said."""
    vals = [float(v) for v in values]
    if dist is None:
        return [vals]
    import torch
    mine = torch.tensor(vals, device=device, dtype=torch.float64)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in t.cpu()] for t in out]
