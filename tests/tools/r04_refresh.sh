# Round-4 artefact refresh (run on the GPU box through tests/tools/gpu.sh, which stamps .build_head).  Usage: bash tests/tools/r04_refresh.sh <part> [tag]
#   part bench  : the full bench line of every BASELINE config (+ c2 with --trsv-block 1024)
#   part trace  : rocprofv3 --kernel-trace --stats of the same commands (side configs and CPU legs off: one config's kernels per trace) + .meta.json
#   part pmc    : HBM traffic per config from two separate PMC passes (never combined with other trace domains) + _meta
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; PART=${1:-bench}; TAG=${2:-r04z}; O=gpurun_out/$TAG; mkdir -p $O
run_bench() { # name, args...
  n=$1; shift
  timeout -k 10 400 python bench.py "$@" > $O/bench_$n.log 2>&1 && tail -1 $O/bench_$n.log > $O/bench_$n.json && python - <<PY
import json; d=json.load(open("$O/bench_$n.json")); p=d.get("parity") or {}
print("$n", d["value"], d["unit"], "roofline", (d.get("roofline") or {}).get("frac"), "parity", p.get("ok"), p.get("max_rel_dev_x"), "side" if "side_configs" in d else "")
PY
  [ $? -ne 0 ] && { echo "bench $n FAILED"; tail -5 $O/bench_$n.log; }
  return 0
}
if [ $PART = bench ]; then
  run_bench c2; run_bench c5 --config c5; run_bench c3 --config c3; run_bench c4 --config c4; run_bench c1 --config c1; run_bench c2_trsv1024 --trsv-block 1024
fi
if [ $PART = trace ]; then
  for spec in "c2:" "c5:--config c5" "c3:--config c3" "c4:--config c4" "c2_trsv1024:--trsv-block 1024"; do
    n=${spec%%:*}; a=${spec#*:}; cmd="python3 bench.py $a --no-cpu-baseline --no-time-to-eps --no-side-configs"
    cd /tmp; export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$n -o bench -- python3 $R/bench.py $a --no-cpu-baseline --no-time-to-eps --no-side-configs > $R/$O/prof_$n.log 2>&1
    rc=$?; cd $R; [ $rc -ge 124 ] && { echo "trace $n timed out: stopping"; exit 1; }
    # the per-dispatch trace summarised by tests/tools/trace_stats_split.py (one row per grid of a kernel, done-flag no-ops apart) instead of rocprofv3's own --stats table
    r=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$r" ] && cp $r $O/bench_${n}_kernel_stats_rocprofv3_raw.csv   # rocprofv3's own --stats table, kept beside the split one
    t=$(find $O/prof_$n -name "*kernel_trace.csv" | head -1)
    [ -n "$t" ] && python tests/tools/trace_stats_split.py $t $O/bench_${n}_kernel_stats.csv && python tests/tools/stats_meta.py $O/bench_${n}_kernel_stats.csv "rocprofv3 --kernel-trace --stats --output-format csv -- $cmd ; tests/tools/trace_stats_split.py" && echo "== $n" && python tests/tools/print_stats.py $O/bench_${n}_kernel_stats.csv 6
    rm -rf $O/prof_$n
  done
fi
if [ $PART = pmc ]; then
  for spec in "c2:" "c5:--config c5" "c3:--config c3" "c4:--config c4" "c2_trsv1024:--trsv-block 1024"; do
    n=${spec%%:*}; a=${spec#*:}; cmd="python3 bench.py $a --no-cpu-baseline --no-time-to-eps --no-side-configs --steps 2 --warmup 0"
    for ctr in FETCH_SIZE WRITE_SIZE; do
      cd /tmp; export TMPDIR=/tmp
      timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/pmc_${ctr}_$n -o p -- python3 $R/bench.py $a --no-cpu-baseline --no-time-to-eps --no-side-configs --steps 2 --warmup 0 > $R/$O/pmc_${ctr}_$n.log 2>&1
      rc=$?; cd $R; [ $rc -ge 124 ] && { echo "pmc $ctr $n timed out: stopping"; exit 1; }
    done
    ff=$(find $O/pmc_FETCH_SIZE_$n -name "*counter_collection.csv" | head -1); fw=$(find $O/pmc_WRITE_SIZE_$n -name "*counter_collection.csv" | head -1)
    [ -n "$ff" ] && [ -n "$fw" ] && python tests/tools/pmc_summary.py $ff $fw $O/pmc_traffic_$n.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- $cmd" > $O/pmc_traffic_$n.txt 2>&1 && echo "== $n" && sed -n 1,4p $O/pmc_traffic_$n.txt
    rm -rf $O/pmc_FETCH_SIZE_$n $O/pmc_WRITE_SIZE_$n
  done
fi
