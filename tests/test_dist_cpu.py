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


def test_balanced_assignments_cover_the_batch_and_even_out_the_work():
    """Runs to a tolerance: the iteration count differs from QP to QP, so equal counts of QPs are not equal work.  Data: the per-QP iteration counts
    of BASELINE config 4's 256 QPs run to eps = 1e-6 on one MI355X (tests/golden/c4_time_to_eps_iterations.json, written by
    tests/tools/gpu_c4_iteration_counts.py: 400 .. 1425 iterations).  Every mode must hand out each QP exactly once; on the recorded counts the ranks'
    summed iterations stay within 10 % of each other for round-robin and within 1 % for LPT -- where LPT is given the recorded counts themselves as its
    work estimate, i.e. its best case.  (Measured spreads at 8 ranks: contiguous 9.8 %, round-robin 9.1 %, LPT 0.8 %: the counts of this set do not drift
    with the index, so round-robin gains little over slabs; the gain is LPT's.)"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, "tests", "golden", "c4_time_to_eps_iterations.json")))
    work = rec["iterations"]
    assert len(work) == 256 and min(work) >= 25 and set(rec["flags"]) == {3}
    for world in (2, 4, 8):
        spread = {}
        for assign in ("contiguous", "round_robin", "lpt"):
            parts = [qd.shard_assign(256, r, world, assign=assign, work=work) for r in range(world)]
            assert sorted(b for p in parts for b in p) == list(range(256)), (assign, world)
            if assign == "contiguous":
                assert parts == [list(range(*qd.shard_range(256, r, world))) for r in range(world)]
            loads = [sum(work[b] for b in p) for p in parts]
            spread[assign] = (max(loads) - min(loads)) / (sum(loads) / world)
        assert spread["round_robin"] < 0.10 and spread["lpt"] < 0.01, (world, spread)
        assert spread["lpt"] <= min(spread["contiguous"], spread["round_robin"]), (world, spread)
    with pytest.raises(ValueError):
        qd.shard_assign(8, 0, 2, assign="lpt")
    with pytest.raises(ValueError):
        qd.shard_assign(8, 0, 2, assign="nope")


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


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) must itself start two ranks: the selftest mode runs the rank plumbing
    (rendezvous on 127.0.0.1, barrier, MAX / SUM reduction, one JSON line from rank 0) without a solver call, so it runs on CPU."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--config", "c1", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_counted"] == 2 and out["steps_counted"] == 6 and out["selftest"] is True
    assert out["config"]["dist_backend"] == "gloo"            # no GPU here: ranks share no card -> gloo


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _run_bench(["--gpus", "2", "--launcher-selftest"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r = _run_bench(["--gpus", "1", "--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
