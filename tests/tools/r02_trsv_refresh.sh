# Artefacts of the single-launch blocked sweeps (run on the GPU box through gpurun): c2 bench line with --trsv-block, kernel stats of the same
# command, HBM traffic of the sweep kernels from two separate PMC passes.  Usage: bash tests/tools/r02_trsv_refresh.sh [tag] [config] [nb]
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r02t}; C=${2:-c2}; NB=${3:-1024}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 400 python bench.py --config $C --trsv-block $NB > $O/bench_${C}_trsv$NB.log 2>&1 && tail -1 $O/bench_${C}_trsv$NB.log > $O/bench_${C}_trsv$NB.json && cut -c1-200 $O/bench_${C}_trsv$NB.json || { echo "bench FAILED"; tail -5 $O/bench_${C}_trsv$NB.log; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python3 $R/bench.py --config $C --trsv-block $NB --no-cpu-baseline --no-time-to-eps > $R/$O/prof.log 2>&1
cd $R; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${C}_trsv${NB}_kernel_stats.csv && python tests/tools/print_stats.py $f 8
cd /tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_f -o f -- python3 $R/bench.py --config $C --trsv-block $NB --no-cpu-baseline --no-time-to-eps --steps 2 --warmup 0 > $R/$O/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_w -o w -- python3 $R/bench.py --config $C --trsv-block $NB --no-cpu-baseline --no-time-to-eps --steps 2 --warmup 0 > $R/$O/pmc_w.log 2>&1
cd $R; ff=$(find $O/pmc_f -name "*counter_collection.csv" | head -1); fw=$(find $O/pmc_w -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tests/tools/pmc_summary.py $ff $fw $O/pmc_traffic_${C}_trsv$NB.json > $O/pmc_traffic_${C}_trsv$NB.txt 2>&1 && sed -n 1,6p $O/pmc_traffic_${C}_trsv$NB.txt
rm -rf $O/pmc_f $O/pmc_w $O/prof
