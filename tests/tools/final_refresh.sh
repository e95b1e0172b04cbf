set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -2 gpurun_out/final/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench.json; cut -c1-300 gpurun_out/final/bench.json
for c in c3 c4 c5; do timeout -k 10 300 python bench.py --config $c > gpurun_out/final/bench_$c.log 2>&1; tail -1 gpurun_out/final/bench_$c.log > gpurun_out/final/bench_$c.json; cut -c1-200 gpurun_out/final/bench_$c.json; done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o bench -- python3 $R/bench.py > $R/gpurun_out/final/prof.log 2>&1
cd $R; f=$(find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1); echo stats=$f; python tests/tools/print_stats.py $f 8
# HBM traffic of the loop kernels: two separate PMC passes (never combined with other trace domains)
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_f -o f -- python3 $R/bench.py --no-cpu-baseline --no-time-to-eps --steps 4 > $R/gpurun_out/final/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_w -o w -- python3 $R/bench.py --no-cpu-baseline --no-time-to-eps --steps 4 > $R/gpurun_out/final/pmc_w.log 2>&1
cd $R; ff=$(find gpurun_out/final/pmc_f -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/final/pmc_w -name "*counter_collection.csv" | head -1)
python tests/tools/pmc_summary.py $ff $fw gpurun_out/final/pmc_traffic.json > gpurun_out/final/pmc_traffic.txt 2>&1; sed -n 1,6p gpurun_out/final/pmc_traffic.txt
