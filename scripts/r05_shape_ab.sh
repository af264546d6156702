#!/bin/bash
: > gpurun_out/r05z_poly3.txt
for case in midpoly3_4 midpoly5_4 midpoly3_12 midpoly5_12 midpoly5_2 smallpoly4_5 poly4 poly3_4 poly12 bin100k_4 ragged4 hiv12; do
  for v in default= old=NO_WIDE_LEAN:1; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_poly3.txt 2>&1
  done
done
cat gpurun_out/r05z_poly3.txt
