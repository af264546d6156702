#!/bin/bash
: > gpurun_out/r05y_shape_final.txt
for case in poly3_64 poly64 mid64 ragged64c16 ragged64 balanced64; do
  for v in default= old=BU_WIDE:1,F81_TD_R:8; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05y_shape_final.txt 2>&1
  done
done
for case in ragged8 ragged12 ragged20 ragged32 poly12 poly20 hiv12; do
  for v in default= old=SORT_LEVELS:0; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05y_shape_final.txt 2>&1
  done
done
cat gpurun_out/r05y_shape_final.txt
