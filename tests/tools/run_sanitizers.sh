#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over every piece of CPU-side code that can be reached without a GPU (round-3 review item 5):
#   * the product's host-only code -- SpMV layout builders + CSC validation / conversion (spmv_layout.cpp), symbolic analysis of the sparse L D L'
#     (ldl_symbolic.cpp) -- built with g++ behind tests/capi/layout_shim.cpp, driven by tests/test_layout_cpu.py;
#   * the CPU oracle (oracle/qps_oracle.c), driven by tests/test_oracle.py and tests/test_polish_oracle.py.
# The sanitizer runtimes are preloaded into the Python process (the libraries are dlopen'ed).  GPU code cannot be sanitised on this pool.
# usage: bash tests/tools/run_sanitizers.sh [logfile]
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
LOG=${1:-gpurun_out/sanitizers.log}; mkdir -p "$(dirname "$LOG")"
make -C quadraticprogramsolver_amd/csrc -s SAN=1 host-test || exit 1
make -C oracle -s san || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
{
  echo "== $(date -u +%FT%TZ) gcc $(gcc -dumpfullversion) -fsanitize=address,undefined -fno-sanitize-recover=undefined; HEAD $(git rev-parse --short HEAD 2>/dev/null)"
  echo "== libraries: quadraticprogramsolver_amd/libqps_host_test_san.so (spmv_layout.cpp, ldl_symbolic.cpp, tests/capi/layout_shim.cpp), oracle/libqps_oracle_san.so (qps_oracle.c)"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 \
    QPS_HOST_TEST_LIB=$PWD/quadraticprogramsolver_amd/libqps_host_test_san.so QPS_ORACLE_LIB=$PWD/oracle/libqps_oracle_san.so \
    python -m pytest tests/test_layout_cpu.py tests/test_oracle.py tests/test_polish_oracle.py -q -x -p no:cacheprovider 2>&1
  echo "== exit code $?"
} | tee "$LOG"
grep -q "== exit code 0" "$LOG"
