"""Phase timeline of k_spmv_blk on BASELINE config 3 from in-kernel s_memtime stamps (diagnostic build: -DQPS_SPMV_STAMPS).
Usage (on the GPU box): bash tests/tools/gpu_c3_stamps.sh   -> rebuilds k_sparse.o with the stamps, runs this script."""
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
with qps.QuadraticProgram(P, q, A, l, u, linsys="cg") as prob:
    x = np.zeros(n)
    prob.solve(x, numIterations=3, ϵAbs=0.0, ϵRel=0.0)
    for rows, name in ((n + m, "[P;A] u"), (n, "A' v")):
        assert L.qps_debug_spmv_stamps(None, 0, rows) == 0          # from now on only launches with this row count are stamped
        x = np.zeros(n)
        prob.solve(x, numIterations=3, ϵAbs=0.0, ϵRel=0.0)
        assert L.qps_debug_spmv_stamps(buf, WGS * SLOTS, rows) == 0
        t = np.frombuffer(buf, dtype=np.int64).reshape(WGS, SLOTS).copy()
        live = t[:, 0] > 0
        t = t[live]
        t0 = t[:, 0].min()
        end = t[:, 63]
        ntask = ((t[:, 3:35:4] > t[:, [0]]).sum(axis=1))
        print(f"== {name}: {live.sum()} workgroups stamped; tasks per workgroup min/mean/max {ntask.min()}/{ntask.mean():.2f}/{ntask.max()}")
        print(f"   kernel span (first start -> last end): {(end.max() - t0) / 100.0:.2f} us at 100 MHz ticks" )
        print(f"   workgroup start offset   mean {np.mean(t[:, 0] - t0) / 100:.2f} us  max {np.max(t[:, 0] - t0) / 100:.2f} us")
        print(f"   start -> meta barrier    mean {np.mean(t[:, 1] - t[:, 0]) / 100:.2f} us")
        print(f"   meta -> x block in LDS   mean {np.mean(t[:, 2] - t[:, 1]) / 100:.2f} us")
        full = ntask >= 2
        for k, lab in ((3, "wait loads + gather + products -> LDS"), (4, "fetch next + barrier 1"), (5, "row sums + stores"), (6, "barrier 2")):
            prev = {3: None, 4: 3, 5: 4, 6: 5}[k]
            d = []
            for ti in range(8):
                sel = ntask > ti
                if not sel.any():
                    break
                a = t[sel, k + 4 * ti]
                b = t[sel, prev + 4 * ti] if prev is not None else (t[sel, 2] if ti == 0 else t[sel, 6 + 4 * (ti - 1)])
                d.append(np.mean(a - b) / 100)
            print(f"   {lab:42s} per task [us]: " + " ".join(f"{v:.2f}" for v in d))
        print(f"   workgroup lifetime        mean {np.mean(end - t[:, 0]) / 100:.2f} us  max {np.max(end - t[:, 0]) / 100:.2f} us")
