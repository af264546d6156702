#!/bin/bash
# Sweep of schedule variants, ONE PROCESS PER VARIANT (scripts/r05_tune_one.py): the A/B method of the second half of round 5.
# usage: r05_tune_sweep.sh "<case> <case> ..." "<name=SWITCH:value,...> <name=...> ..." [output file under gpurun_out/]
# e.g.   r05_tune_sweep.sh "poly3_20 ragged20" "default= off=NO_THIN:1 t1k=THIN_UNITS:1024"
# PASTML_HIP_LIBRARY=<other build> in the environment compares libraries the same way.
OUT=gpurun_out/${3:-r05_tune_sweep.txt}
mkdir -p gpurun_out
: > $OUT
for case in $1; do
  for v in $2; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> $OUT 2>&1
  done
done
cat $OUT
