"""N > 1 path on CPU: world_size-2 gloo rendezvous exercising the sharding and the timing reduction bench.py uses."""
import multiprocessing as mp
import os

import pytest

from quadraticprogramsolver_amd import dist as qd


def test_shard_ranges_tile_the_batch():
    for total in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [qd.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert qd.shard_range(256, 3, 8) == (96, 128)     # BASELINE config 4: QP b -> GPU b // 32


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    info = qd.rank_info_from_env()
    backend = qd.init_process_group(info, prefer="gloo")
    begin, end = qd.shard_range(10, info.rank, info.world_size)
    qd.barrier(info)
    elapsed = 1.0 + rank                 # rank 1 is the slow one
    units = (end - begin) * 100.0        # e.g. iterations processed by this rank
    value, tmax = qd.gather_timings(info, elapsed, units)
    (s,) = qd.all_sum(info, end - begin)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, backend, value, tmax, s))


def test_two_rank_gloo_timing_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, backend, value, tmax, s in res:
        assert backend == "gloo"
        assert tmax == 2.0                   # MAX over ranks
        assert s == 10.0                     # the two slabs cover the batch
        assert value == pytest.approx(1000.0 / 2.0)   # whole-job units / slowest rank
