# C3 counter diagnosis (round-2 review item 3): one rocprofv3 --pmc pass per counter group over `bench.py --config c3` (program directly after
# `--`, --kernel-trace only).  Usage (through gpurun): bash tests/tools/r03_c3_counters.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r03_c3pmc}; O=gpurun_out/$TAG; mkdir -p $O
i=0; files=""
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1)); cd /tmp; export TMPDIR=/tmp
  timeout -k 10 240 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $R/$O/p$i -o p -- python3 $R/bench.py --config c3 --no-cpu-baseline --steps 2 --warmup 0 > $R/$O/p$i.log 2>&1
  rc=$?; cd $R
  if [ $rc -ge 124 ]; then echo "pass $i ($group) timed out: stopping"; break; fi
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/pass${i}_counters.csv; files="$files $O/pass${i}_counters.csv"; echo "pass $i ok: $group"; else echo "pass $i ($group) produced no counters (rc $rc)"; tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TD_TD_BUSY_sum TA_TA_BUSY_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
python tests/tools/pmc_counters_summary.py $O/c3_counters.json $files > $O/c3_counters.txt 2>&1; cut -c1-1500 $O/c3_counters.txt; rm -f $O/pass*_counters.csv
