#!/bin/bash
# kernel stats + HBM counters + SQ counters of the ragged 262 144-tip marginal pass (k = 64, 32 characters), for the
# library in place and, if given, for scratch/$2 as well.  $1 = case (ragged64), output under gpurun_out/prof_<tag>_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
CASE=${1:-ragged64}
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
run() {
  tag=$1
  rm -rf $O/prof_${tag}_*
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_kt -o run -- python3 $R/scripts/prof_forest_driver.py $CASE 5 > $O/prof_${tag}_kt.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_${tag}_fetch -o run -- python3 $R/scripts/prof_forest_driver.py $CASE 1 > $O/prof_${tag}_fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_${tag}_write -o run -- python3 $R/scripts/prof_forest_driver.py $CASE 1 > $O/prof_${tag}_write.log 2>&1 || return 1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/prof_${tag}_sqa -o run -- python3 $R/scripts/prof_forest_driver.py $CASE 1 > $O/prof_${tag}_sqa.log 2>&1 || return 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/prof_${tag}_sqb -o run -- python3 $R/scripts/prof_forest_driver.py $CASE 1 > $O/prof_${tag}_sqb.log 2>&1 || return 1
  python3 $R/scripts/prof_forest_summary.py $O/prof_${tag}_fetch $O/prof_${tag}_write $O/prof_${tag}_sqa $O/prof_${tag}_sqb > $O/prof_${tag}_summary.txt
  find $O/prof_${tag}_kt -name '*kernel_stats.csv' -exec cp {} $O/prof_${tag}_kernel_stats.csv \;
  # keep the merge-back small: the per-dispatch csv files are large
  find $O/prof_${tag}_* -name '*.csv' -size +2M -delete
}
run A_$CASE || { echo "profile A failed"; tail -5 $O/prof_A_${CASE}_*.log; exit 1; }
if [ -n "$2" ]; then
  export PASTML_HIP_LIBRARY=$R/scratch/$2
  run B_$CASE || { echo "profile B failed"; unset PASTML_HIP_LIBRARY; exit 1; }
  unset PASTML_HIP_LIBRARY
fi
cat $O/prof_*_${CASE}_summary.txt
