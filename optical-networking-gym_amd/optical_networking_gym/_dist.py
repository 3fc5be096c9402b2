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


def allreduce_mean_gradients(params, dist=None):
    """Data-parallel learner step over env shards (BASELINE config 5: the reference feeds ONE MaskablePPO learner from 14
    SubprocVecEnv workers, examples/ONDM_2025/train_multi_masked_ppo.py:410-458; here every rank steps its own shard of
    the replicas and owns a copy of the learner): the gradients of all parameters travel in ONE flat bucket through ONE
    all-reduce (RCCL over xGMI on the GPUs, gloo in the CPU tests) and are averaged over the ranks, so every rank applies
    the same update.  xGMI rings are per-link bound: one ~20 MB bucket instead of one collective per tensor."""
    params = [p for p in params if p.grad is not None]
    if dist is None or not params:
        return
    import torch
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
