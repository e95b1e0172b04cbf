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


def shard_assign(total: int, rank: int, world_size: int, assign: str = "contiguous", work=None):
    """Indices of the independent QPs owned by `rank` ("balance by assigning QPs round-robin or by work-stealing").

    ``contiguous``  the slabs of `shard_range` (QP b -> rank b // ceil(total / world)): what the fixed-K throughput line uses -- every QP runs the
                    same number of iterations there, so equal counts are equal work, and a slab is one batched handle.
    ``round_robin`` QP b -> rank b % world: for runs to a tolerance, where the iteration count differs from QP to QP (BASELINE config 4: 425-975
                    iterations inside one slab) and neighbouring QPs of a generated set tend to resemble each other.
    ``lpt``         longest-processing-time-first on a per-QP work estimate `work[b]` (e.g. the iteration counts of a previous solve of the same set,
                    or of a loose-tolerance pre-solve): QPs by decreasing work, each to the rank with the least work so far; ties by index, so
                    every rank computes the same assignment without talking to the others.
    No data-path collective in any mode: a rank only needs to know its own list."""
    if assign == "contiguous":
        begin, end = shard_range(total, rank, world_size)
        return list(range(begin, end))
    if assign == "round_robin":
        return list(range(rank, total, world_size))
    if assign == "lpt":
        if work is None or len(work) != total:
            raise ValueError("assign='lpt' needs a work estimate per QP")
        load = [0.0] * world_size
        mine = []
        for b in sorted(range(total), key=lambda i: (-float(work[i]), i)):
            r = min(range(world_size), key=lambda k: (load[k], k))
            load[r] += float(work[b])
            if r == rank:
                mine.append(b)
        return sorted(mine)
    raise ValueError(f"unknown assignment {assign!r}")


def handout_ranges(total: int, workers: int, chunk: int):
    """The ranges qps_solve_batch_multi cuts a batch into (quadraticprogramsolver_amd/csrc/batch_schedule.h), in hand-out order: chunk > 0 -- guided self-scheduling,
    take(b) = clamp(ceil(remaining / (2 W)), max(1, chunk / 4), chunk); chunk <= 0 -- one contiguous slab per worker."""
    if chunk <= 0:
        slab = max(1, -(-total // workers))
        return [(b, min(slab, total - b)) for b in range(0, total, slab)]
    out, b, smallest = [], 0, max(1, chunk // 4)
    while b < total:
        rem = total - b
        k = min(rem, max(smallest, min(chunk, -(-rem // (2 * workers)))))
        out.append((b, k)); b += k
    return out


def handout_schedule(work, workers: int, chunk: int, lockstep: bool = False):
    """Event simulation of the in-process hand-out: the ranges of `handout_ranges`, each to the worker that becomes free first (ties: lowest index; static slabs:
    worker = slab index modulo workers).  Cost of a range: the sum of its QPs' work, or -- ``lockstep`` -- (its longest QP) x (its size): a batched handle advances
    its QPs together, and a small range is launch-bound rather than bandwidth-bound.  Returns (worker of every QP, load per worker)."""
    import heapq
    total = len(work)
    cost = (lambda v: max(v) * len(v)) if lockstep else sum
    load = [0.0] * workers
    owner = [0] * total
    free = [(0.0, w) for w in range(workers)]
    heapq.heapify(free)
    for i, (b, k) in enumerate(handout_ranges(total, workers, chunk)):
        if chunk <= 0:
            w, t = i % workers, load[i % workers]
        else:
            t, w = heapq.heappop(free)
        c = float(cost([float(x) for x in work[b:b + k]]))
        load[w] += c
        if chunk > 0:
            heapq.heappush(free, (t + c, w))
        for j in range(b, b + k):
            owner[j] = w
    return owner, load


def init_process_group(info: RankInfo, prefer: str | None = None):
    """Returns the backend actually in use ("nccl" == RCCL on ROCm, or "gloo"), or None for a single process.

    The ADMM path has no exchange step: the only traffic is the barrier around a timed region and a MAX / SUM of a few scalars.
    `prefer` (bench.py: "nccl" when every rank owns a GPU, "gloo" when ranks share a card or there is none; QPS_DIST_BACKEND
    overrides) is honoured when possible; RCCL needs one distinct device per rank.  A single process creates no group unless
    QPS_DIST_FORCE_GROUP=1 (used to exercise the RCCL initialisation and all-reduce on a 1-GPU box)."""
    prefer = prefer or os.environ.get("QPS_DIST_BACKEND", "gloo")
    force = os.environ.get("QPS_DIST_FORCE_GROUP") == "1"
    if info.world_size <= 1 and not force:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        return dist.get_backend()
    ndev = torch.cuda.device_count()
    backend = prefer if (prefer == "gloo" or (ndev >= info.world_size and ndev > 0)) else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(info.local_rank % ndev)
        dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world_size, device_id=torch.device("cuda", info.local_rank % ndev))
    else:
        dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world_size)
    return backend


def _grouped():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def shutdown(info: RankInfo):
    if _grouped():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def barrier(info: RankInfo):
    if _grouped():
        import torch.distributed as dist
        dist.barrier()


def _reduce(info: RankInfo, values, op_name: str):
    if not _grouped():
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
