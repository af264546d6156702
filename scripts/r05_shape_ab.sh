#!/bin/bash
: > gpurun_out/r05z_r2.txt
for case in poly20 poly3_20 poly3_32 ragged20 mid64; do
  for v in default= r2=F81_R:2,F81_TD_R:2 td2=F81_TD_R:2 bu2=F81_R:2; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_r2.txt 2>&1
  done
done
cat gpurun_out/r05z_r2.txt
