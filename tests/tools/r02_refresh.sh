# Round-2 artefact refresh (run on the GPU box through gpurun): bench line per BASELINE config, kernel trace of the headline run, HBM
# traffic per config from two separate PMC passes (never combined with other trace domains).  Usage: bash tests/tools/r02_refresh.sh [tag] [configs...]
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r02}; shift; CFGS=${@:-c2 c5 c3 c4 c1}
O=gpurun_out/$TAG; mkdir -p $O
for c in $CFGS; do
  timeout -k 10 400 python bench.py --config $c > $O/bench_$c.log 2>&1 && tail -1 $O/bench_$c.log > $O/bench_$c.json && cut -c1-260 $O/bench_$c.json || { echo "bench $c FAILED"; tail -5 $O/bench_$c.log; }
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_c2_kernel_stats.csv && python tests/tools/print_stats.py $f 10
for c in $CFGS; do
  [ $c = c1 ] && continue
  cd /tmp
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_f_$c -o f -- python3 $R/bench.py --config $c --no-cpu-baseline --no-time-to-eps --steps 2 --warmup 0 > $R/$O/pmc_f_$c.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_w_$c -o w -- python3 $R/bench.py --config $c --no-cpu-baseline --no-time-to-eps --steps 2 --warmup 0 > $R/$O/pmc_w_$c.log 2>&1
  cd $R; ff=$(find $O/pmc_f_$c -name "*counter_collection.csv" | head -1); fw=$(find $O/pmc_w_$c -name "*counter_collection.csv" | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python tests/tools/pmc_summary.py $ff $fw $O/pmc_traffic_$c.json > $O/pmc_traffic_$c.txt 2>&1 && sed -n 1,4p $O/pmc_traffic_$c.txt
  rm -rf $O/pmc_f_$c $O/pmc_w_$c
done
rm -rf $O/prof
