cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03e
for v in "QPS_SPMV_WGS=504 QPS_SPMV_JDS=0" "QPS_SPMV_WGS=504 QPS_SPMV_JDS=1" "QPS_SPMV_WGS=448 QPS_SPMV_JDS=0" "QPS_SPMV_WGS=512 QPS_SPMV_JDS=0"; do
  tag=$(echo $v | tr ' =' '__')
  env $v timeout -k 10 200 python bench.py --config c3 --no-cpu-baseline --steps 5 > gpurun_out/r03e/$tag.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03e/$tag.log").read().strip().splitlines()[-1])
print("$v", d["value"], d["cg_iterations_per_s"], [(k["name"][:16], k["avg_us"]) for k in d["kernels"]])
PY
done
