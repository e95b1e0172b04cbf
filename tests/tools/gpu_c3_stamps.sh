# Diagnostic build with in-kernel stamps + the timeline script (run through gpurun).  The snapshot on the box is scratch: the product .so is not touched here.
set -e
cd $GRAFT_REPO_ROOT/quadraticprogramsolver_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-result -DQPS_SPMV_STAMPS -c k_sparse.hip -o k_sparse.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libqps_hip.so qps_capi.o qps_proxqp.o k_loop.o k_pass.o k_pass_pq.o k_trsv.o k_trsv_blocked.o k_small.o k_setup.o k_sparse.o qps_polish.o k_ldl.o ldl_symbolic.o
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f
timeout -k 10 300 python tests/tools/gpu_c3_stamps.py > gpurun_out/r03f/c3_stamps.txt 2>&1; cat gpurun_out/r03f/c3_stamps.txt
