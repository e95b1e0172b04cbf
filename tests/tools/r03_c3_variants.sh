# c3 bench under a few environment variants (run through gpurun): bash tests/tools/r03_c3_variants.sh "VAR=1 VAR2=2" "..." 
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03e
for v in "$@"; do
  tag=$(echo $v | tr ' =' '__')
  env $v timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline --steps 5 > gpurun_out/r03e/$tag.log 2>&1 || { tail -5 gpurun_out/r03e/$tag.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03e/$tag.log").read().strip().splitlines()[-1])
print("$v", d["value"], d["cg_iterations_per_s"], [(k["name"][:16], k["avg_us"]) for k in d["kernels"]])
PY
done
