"""Multi-GPU plumbing: one process per GPU, no data-path collective.

The ADMM path shards only across independent QPs: a batch is cut into contiguous slabs, one per rank,
and each rank runs its own device-resident loop.  torch.distributed (RCCL on GPUs, gloo on CPU) is used for exactly two
things: the barrier that brackets a timed region and the MAX/SUM reduction of a few scalars (timings, iteration counts).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class RankInfo:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1


def rank_info_from_env() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                    int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total: int, rank: int, world_size: int):
    """Contiguous slab [begin, end) of `total` independent QPs owned by `rank` (QP b -> rank b // ceil(total/world)):
    sizes differ by at most one and the slabs tile [0, total) without overlap."""
    base, extra = divmod(total, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def init_process_group(info: RankInfo, prefer: str | None = None):
    """Returns the backend actually in use ("gloo", or "nccl" == RCCL on ROCm), or None for a single process.

    Default is gloo: the ADMM path has no exchange step, the only traffic is a barrier and two scalars per timed
    region, and keeping torch off the GPU leaves a single HIP context per rank (the solver's).  QPS_DIST_BACKEND=nccl
    (or prefer="nccl") routes those scalars through RCCL instead."""
    prefer = prefer or os.environ.get("QPS_DIST_BACKEND", "gloo")
    if info.world_size <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        return dist.get_backend()
    backend = prefer if (prefer == "gloo" or torch.cuda.is_available()) else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(info.local_rank)
    dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world_size)
    return backend


def barrier(info: RankInfo):
    if info.world_size > 1:
        import torch.distributed as dist
        dist.barrier()


def _reduce(info: RankInfo, values, op_name: str):
    if info.world_size <= 1:
        return [float(v) for v in values]
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return [float(v) for v in t.cpu().tolist()]


def all_max(info: RankInfo, *values):
    return _reduce(info, values, "MAX")


def all_sum(info: RankInfo, *values):
    return _reduce(info, values, "SUM")


def gather_timings(info: RankInfo, elapsed_s: float, units_done: float):
    """Whole-job throughput of a weak-scaling run: (sum over ranks of the units each processed) / (max elapsed)."""
    (tmax,) = all_max(info, elapsed_s)
    (units,) = all_sum(info, units_done)
    return units / tmax, tmax
