#!/bin/bash
# Sparse ProxQP with G x of the row updates from the product (default) against QPS_PROXQP_GX_SOLVE=1 (from the KKT solve): tests, fuzz seeds, timing.
set -o pipefail
O=gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_proxqp.py -m gpu -x -q > $O/r04_zz_pytest_proxqp.log 2>&1 || { tail -20 $O/r04_zz_pytest_proxqp.log; exit 1; }
tail -n 1 $O/r04_zz_pytest_proxqp.log
for sd in 43 23 29; do
  timeout -k 10 120 python tests/tools/gpu_fuzz_proxqp.py $([ $sd = 43 ] && echo 300 || echo 200) $sd > $O/r04_zz_fuzz_proxqp_seed${sd}_gx_product.log 2>&1 || exit 2
  tail -n 1 $O/r04_zz_fuzz_proxqp_seed${sd}_gx_product.log
done
: > $O/r04_zz_proxqp_sparse_gx_timing.log
for gx in 0 1; do
  for args in "2000 400 3000 0.004" "8000 1000 12000 0.0005" "200000 0 0 -1"; do
    QPS_PROXQP_GX_SOLVE=$gx timeout -k 10 120 python tests/tools/gpu_proxqp_sparse_timing.py $args 2>&1 | cut -c1-260 | sed "s/^/GX_SOLVE=$gx /" >> $O/r04_zz_proxqp_sparse_gx_timing.log || exit 3
  done
done
cat $O/r04_zz_proxqp_sparse_gx_timing.log | cut -c1-200
