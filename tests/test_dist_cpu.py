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


def _c4_counts():
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return json.load(open(os.path.join(root, "tests", "golden", "c4_time_to_eps_iterations.json")))["iterations"]


def test_dynamic_handout_balances_a_run_to_a_tolerance():
    """The in-process multi-device driver (qps_solve_batch_multi, batch_schedule.h; SURVEY 8e "work-stealing at chunk boundaries") on the recorded per-QP iteration
    counts of BASELINE config 4: ranges of at most `chunk` QPs, shrinking towards the end of the batch, each to the worker that frees first.  The ranges tile the batch;
    at 8 workers and chunk = 8 the makespan stays within 3 % of the mean load under both cost models (bandwidth-bound: the sum of a range's iterations; lock step: its
    longest QP x its size) -- where the static contiguous slabs are 5-10 % off and need no knowledge either, and LPT's 0.8 % needs the answer in advance."""
    work = _c4_counts()
    for workers in (2, 4, 8):
        for chunk in (8, 16, 32):
            ranges = qd.handout_ranges(256, workers, chunk)
            assert ranges[0] == (0, min(chunk, max(chunk // 4, -(-256 // (2 * workers)))))
            assert all(a + k == b for (a, k), (b, _) in zip(ranges[:-1], ranges[1:])) and ranges[-1][0] + ranges[-1][1] == 256
            assert all(1 <= k <= chunk for _, k in ranges)
    over = {}
    for lockstep in (False, True):
        owner, load = qd.handout_schedule(work, 8, 8, lockstep=lockstep)
        assert sorted(set(owner)) == list(range(8))
        over[lockstep] = max(load) / (sum(load) / 8) - 1.0
    assert over[False] <= 0.03 and over[True] <= 0.03, over
    _, static = qd.handout_schedule(work, 8, 0)
    assert max(static) / (sum(static) / 8) - 1.0 > over[False]                  # the slabs of a fixed-K run are worse on a run to a tolerance
    assert qd.handout_ranges(256, 8, 0) == [(32 * w, 32) for w in range(8)]     # chunk <= 0: QP b -> worker b // 32, as bench.py's slabs


def test_the_handout_itself_with_a_stand_in_for_the_solve():
    """batch_schedule.h as the library runs it -- real host threads, a shared counter -- through the host test library (tests/capi/layout_shim.cpp), the solve replaced by
    a sleep proportional to the recorded iteration counts: every QP solved exactly once, the ranges are those of `handout_ranges` whoever took them, the loads (from the
    recorded counts, not from the wall clock) stay within 8 % of the mean at 8 workers, an error in one range stops the hand-out, chunk <= 0 gives the static slabs."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("QPS_HOST_TEST_LIB")
    if not path:
        subprocess.check_call(["make", "-C", os.path.join(root, "quadraticprogramsolver_amd", "csrc"), "-s", "host-test"])
        path = os.path.join(root, "quadraticprogramsolver_amd", "libqps_host_test.so")
    L = C.CDLL(path)
    L.lt_schedule.argtypes = [C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int64]
    work = _c4_counts()
    cost = (C.c_double * 256)(*[w * 2.0 for w in work])                          # 2 us per iteration: ~0.34 s of sleep in all
    for workers, chunk in ((8, 8), (3, 16), (8, 0)):
        owner, solved, secs = (C.c_int * 256)(), (C.c_int * 256)(), (C.c_double * workers)()
        assert L.lt_schedule(256, workers, chunk, cost, owner, solved, secs, -1) == 0
        assert list(solved) == [1] * 256
        for b, k in qd.handout_ranges(256, workers, chunk):                      # a range is solved by ONE worker, whoever it was
            assert len(set(owner[b:b + k])) == 1, (workers, chunk, b, k)
        load = [sum(work[b] for b in range(256) if owner[b] == w) for w in range(workers)]
        if chunk > 0:
            assert max(load) / (sum(load) / workers) - 1.0 <= 0.08, (workers, chunk, load)
        else:
            assert list(owner) == [b // 32 for b in range(256)]
    owner, solved, secs = (C.c_int * 256)(), (C.c_int * 256)(), (C.c_double * 4)()
    assert L.lt_schedule(256, 4, 8, cost, owner, solved, secs, 40) == 7           # the range holding QP 40 fails: its code comes back ...
    assert sum(solved) < 256                                                     # ... and the hand-out stopped early


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
