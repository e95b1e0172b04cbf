# Kernel-trace summaries of the side configs + multi-rank rehearsals on the 1-GPU box (run through gpurun).  Usage: bash tests/tools/r02_stats_refresh.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r02u}; O=gpurun_out/$TAG; mkdir -p $O
for c in c3 c4 c5; do
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o bench -- python3 $R/bench.py --config $c --no-cpu-baseline --no-time-to-eps > $R/$O/prof_$c.log 2>&1
  cd $R; f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_${c}_kernel_stats.csv && echo "== $c" && python tests/tools/print_stats.py $f 6
  rm -rf $O/prof_$c
done
timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-time-to-eps > $O/bench_c2_2ranks.log 2>&1 && tail -1 $O/bench_c2_2ranks.log > $O/bench_c2_2ranks_on_1gpu.json && cut -c1-220 $O/bench_c2_2ranks_on_1gpu.json
timeout -k 10 300 python bench.py --config c4 --gpus 4 --no-cpu-baseline > $O/bench_c4_4ranks.log 2>&1 && tail -1 $O/bench_c4_4ranks.log > $O/bench_c4_4ranks_on_1gpu.json && cut -c1-220 $O/bench_c4_4ranks_on_1gpu.json
QPS_DIST_FORCE_GROUP=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-time-to-eps --steps 3 > $O/bench_c2_rccl.log 2>&1 && tail -1 $O/bench_c2_rccl.log > $O/bench_c2_rccl_group_1rank.json && python -c "import json,sys; d=json.loads(open('$O/bench_c2_rccl_group_1rank.json').read()); print(d['value'], d['config'].get('dist_backend'))"
