"""
Multi-GPU plumbing for the sweep (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
on CPU for tests).

The path shards over INDEPENDENT problems for every D that fits one workgroup ("replicas only"
below D ~ 256): each rank owns a contiguous slice of the problem list and there is no data-path collective.  The
only exchanges are scalars: the per-problem objective values (all_gather) and the timing reduction of the
benchmark (all_reduce MAX).
"""
import os

import numpy as np


def shard_range(n_problems: int, rank: int, world: int):
    """Contiguous, balanced slice [lo, hi) of `n_problems` owned by `rank` (first ranks get the remainder)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(n_problems, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: str, device_index=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (127.0.0.1 by default).  Returns (rank, world)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl" and device_index is not None:
            import torch
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


def _tensor(values, device):
    import torch
    return torch.as_tensor(np.asarray(values, dtype=np.float64), device=device)


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX all-reduce of one scalar (the benchmark's elapsed time)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = _tensor([value], device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objective(f_local, n_problems: int, device="cpu") -> np.ndarray:
    """All ranks receive the objective of every problem, in problem order (ranks own `shard_range` slices)."""
    import torch
    import torch.distributed as dist
    f_local = np.atleast_1d(np.asarray(f_local, dtype=np.float64))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return f_local.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_problems, rank, world)
    if f_local.size != hi - lo:
        raise ValueError(f"rank {rank} owns {hi - lo} problems but passed {f_local.size} values")
    width = (n_problems + world - 1) // world          # equal-sized slots for all_gather
    buf = torch.zeros(width, dtype=torch.float64, device=device)
    buf[: f_local.size] = _tensor(f_local, device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = np.empty(n_problems)
    for r in range(world):
        a, b = shard_range(n_problems, r, world)
        res[a:b] = out[r][: b - a].cpu().numpy()
    return res


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
