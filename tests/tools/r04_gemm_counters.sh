# fp32 GEMM counter diagnosis: one rocprofv3 --pmc pass per counter group over tests/tools/micro/gemm_f32_bench (program directly after `--`, --kernel-trace only).
# Usage (through gpurun): bash tests/tools/r04_gemm_counters.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r04_gemmpmc}; O=gpurun_out/$TAG; mkdir -p $O
$R/tests/tools/micro/gemm_f32_bench 20 > $O/timing.log 2>&1; cat $O/timing.log
i=0; files=""
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1)); cd /tmp; export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $R/$O/p$i -o p -- $R/tests/tools/micro/gemm_f32_bench 3 > $R/$O/p$i.log 2>&1
  rc=$?; cd $R
  if [ $rc -ge 124 ]; then echo "pass $i ($group) timed out: stopping"; break; fi
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/pass${i}_counters.csv; files="$files $O/pass${i}_counters.csv"; echo "pass $i ok: $group"; else echo "pass $i ($group) produced no counters (rc $rc)"; tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
python tests/tools/pmc_counters_summary.py $O/gemm_counters.json $files > $O/gemm_counters.txt 2>&1; cut -c1-2200 $O/gemm_counters.txt; rm -f $O/pass*_counters.csv
