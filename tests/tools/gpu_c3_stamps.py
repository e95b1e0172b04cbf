"""Phase timeline of the column-blocked SpMV on BASELINE config 3 from in-kernel s_memtime stamps (diagnostic build: -DQPS_SPMV_STAMPS).
Usage (on the GPU box): bash tests/tools/gpu_c3_stamps.sh   -> rebuilds k_sparse.o with the stamps, runs this script.
Stamps (thread 0 of every workgroup, i.e. wave 0): 0 = start, 1 = x block arrived in registers, 2 = x block in LDS (barrier), 3 + k = k-th unit of work of
wave 0 done (a slice of the sliced form / a phase of a task of the task form), 63 = end.  s_memtime ticks are shader cycles."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadraticprogramsolver_amd as qps
from quadraticprogramsolver_amd import _lib

n, m = 50000, 100000
P, q, A, l, u = qps.GenerateSparseBenchmarkQP(n, m, seed=1234)
L = _lib.lib()
L.qps_debug_spmv_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int, C.c_int]
WGS, SLOTS = 1024, 64
buf = (C.c_longlong * (WGS * SLOTS))()
GHZ = float(os.environ.get("QPS_STAMP_GHZ", "2.1"))
with qps.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
    x = np.zeros(n)
    prob.solve(x, numIterations=3, ϵAbs=0.0, ϵRel=0.0)
    for rows, name in ((n + m, "[P;A] u"), (n, "A' v")):
        assert L.qps_debug_spmv_stamps(None, 0, rows) == 0          # from now on only launches with this row count are stamped
        assert L.qps_debug_spmv_stamps(None, -1, rows) == 0         # clear
        x = np.zeros(n)
        prob.solve(x, numIterations=1, ϵAbs=0.0, ϵRel=0.0, numItrPcg=3)     # the table holds the LAST stamped launch
        assert L.qps_debug_spmv_stamps(buf, WGS * SLOTS, rows) == 0
        t = np.frombuffer(buf, dtype=np.int64).reshape(WGS, SLOTS).copy()
        live = (t[:, 0] > 0) & (t[:, 63] > 0)
        t = t[live]
        t0 = t[:, 0].min()
        us = lambda c: c / (GHZ * 1e3)
        end = t[:, 63]
        print(f"== {name}: {live.sum()} workgroups; launch span (first start -> last end) {us(end.max() - t0):.2f} us at {GHZ} GHz")
        so = np.sort(t[:, 0] - t0)
        print(f"   workgroup start offset [us]: median {us(np.median(so)):.2f}  p90 {us(so[int(0.9 * len(so))]):.2f}  max {us(so[-1]):.2f}")
        print(f"   start -> x in registers     mean {us(np.mean(t[:, 1] - t[:, 0])):.2f} us   max {us(np.max(t[:, 1] - t[:, 0])):.2f}")
        print(f"   x in registers -> barrier   mean {us(np.mean(t[:, 2] - t[:, 1])):.2f} us")
        work = t[:, 3:63]
        cnt = (work > 0).sum(axis=1)
        print(f"   units of work of wave 0 per workgroup: min {cnt.min()} mean {cnt.mean():.2f} max {cnt.max()}")
        per = []
        for k in range(int(cnt.max())):
            sel = cnt > k
            prev = t[sel, 2] if k == 0 else t[sel, 3 + k - 1]
            per.append(us(np.mean(t[sel, 3 + k] - prev)))
        print("   duration of unit k [us]: " + " ".join(f"{v:.2f}" for v in per))
        life = end - t[:, 0]
        print(f"   workgroup lifetime [us]: mean {us(life.mean()):.2f}  p90 {us(np.sort(life)[int(0.9 * len(life))]):.2f}  max {us(life.max()):.2f};  end offset of the last-finishing workgroup {us(end.max() - t0):.2f}, of the median {us(np.median(end) - t0):.2f}")
