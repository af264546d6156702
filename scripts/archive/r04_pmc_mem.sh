#!/bin/bash
# memory-system counters of one marginal pass: $1 = case of r04_prof_driver.py, $2 = tag, rest = env settings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
CASE=$1; TAG=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_BUSY_avr TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  rm -rf $O/r04mem_${TAG}_$i
  env "$@" rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/r04mem_${TAG}_$i -o run -- python3 $R/scripts/r04_prof_driver.py $CASE 1 > $O/r04mem_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/r04mem_${TAG}_$i.log; }
done
python3 $R/scripts/r04_prof_summary.py $O/r04mem_${TAG}_1 $O/r04mem_${TAG}_2 $O/r04mem_${TAG}_3 $O/r04mem_${TAG}_4 $O/r04mem_${TAG}_5 $O/r04mem_${TAG}_6 | grep "==\|f81" > $O/r04mem_${TAG}.txt
rm -rf $O/r04mem_${TAG}_[1-6]
cat $O/r04mem_${TAG}.txt
