#!/bin/bash
: > gpurun_out/r05y_shape_ab7.txt
for case in ragged8 ragged16 ragged32 poly12 ragged12 balanced12; do
  for v in default= sort=SORT_LEVELS:1 default= sort=SORT_LEVELS:1; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05y_shape_ab7.txt 2>&1
  done
done
cat gpurun_out/r05y_shape_ab7.txt
