#!/bin/bash
: > gpurun_out/r05z_small_td.txt
for case in hiv40 hiv64 small40 mid64; do
  for v in default= td4=F81_TD_R:4 default= td4=F81_TD_R:4; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_small_td.txt 2>&1
  done
done
cat gpurun_out/r05z_small_td.txt
