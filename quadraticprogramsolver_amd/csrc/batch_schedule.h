// batch_schedule.h -- how a batch of independent QPs is dealt to the workers of ONE process (qps_solve_batch_multi).  Plain C++, header only: the host test library
// (tests/capi/layout_shim.cpp) runs the same code with a stand-in for the solve.
//
// Reference: the per-problem loop of RunBenchmarks.jl:88-104; SURVEY 8e: "one host thread + one stream per device", "round-robin or work-stealing at chunk boundaries".
//   chunk > 0 : every worker repeatedly takes the next range of problems from a shared counter until none are left.  A range starts at `chunk` problems and shrinks
//               towards the end of the batch (guided self-scheduling): take(b) = clamp(ceil(remaining / (2 W)), max(1, chunk / 4), chunk) -- large ranges while there
//               is plenty of work (a batched handle is more efficient the more QPs it carries), small ones for the tail, where the last ranges decide the makespan.
//               A run to a tolerance, in which every QP stops at its own iteration, balances itself: on the recorded iteration counts of BASELINE config 4 (425-975
//               iterations per QP) 8 workers with chunk = 8 finish within 2.5 % of the mean, against 9.8 % for contiguous slabs (tests/test_dist_cpu.py);
//   chunk <= 0: static contiguous slabs, worker w takes [w * slab, (w + 1) * slab) with slab = ceil(count / workers) (capped at max_range; further slabs round-robin):
//               what a fixed-K run wants.
// The size of a range is a function of where it starts, so the ranges are cut the same way whoever solves them: the results cannot depend on the number of worker
// threads' timing (they do depend on `chunk` and on the worker COUNT, which enter the cut).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

namespace qps {

// solve(worker, first, count) -> 0 on success; a non-zero return stops the hand-out (the other workers finish the range they hold) and is returned.
// The calling thread is worker 0.  worker_seconds (may be null): busy time of every worker.
template <typename SolveRange>
int run_batch_workers(int64_t count, int num_workers, int64_t chunk, int64_t max_range, SolveRange&& solve, double* worker_seconds) {
    const int64_t slab = chunk > 0 ? std::min(chunk, max_range) : std::max<int64_t>(1, std::min<int64_t>((count + num_workers - 1) / num_workers, max_range));
    const int64_t smallest = std::max<int64_t>(1, slab / 4);
    auto take_at = [&](int64_t b0) {                                               // size of the range that starts at b0 (chunk > 0)
        const int64_t rem = count - b0, guided = (rem + 2 * (int64_t)num_workers - 1) / (2 * (int64_t)num_workers);
        return std::min(rem, std::max(smallest, std::min(slab, guided)));
    };
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto worker = [&](int w) {
        const double t0 = now();
        if (chunk > 0) {
            for (;;) {
                if (failed.load()) break;
                int64_t b0 = next.load(), k = 0;
                do { if (b0 >= count) break; k = take_at(b0); } while (!next.compare_exchange_weak(b0, b0 + k));
                if (b0 >= count) break;
                const int rc = solve(w, b0, k);
                if (rc != 0) { int expect = 0; failed.compare_exchange_strong(expect, rc); break; }
            }
        } else {
            for (int64_t b0 = (int64_t)w * slab; b0 < count && !failed.load(); b0 += (int64_t)num_workers * slab) {
                const int rc = solve(w, b0, std::min(slab, count - b0));
                if (rc != 0) { int expect = 0; failed.compare_exchange_strong(expect, rc); break; }
            }
        }
        if (worker_seconds) worker_seconds[w] = now() - t0;
    };
    std::vector<std::thread> threads;
    for (int w = 1; w < num_workers; ++w) threads.emplace_back(worker, w);
    worker(0);
    for (auto& t : threads) t.join();
    return failed.load();
}

}  // namespace qps
