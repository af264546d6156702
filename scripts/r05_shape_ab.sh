#!/bin/bash
: > gpurun_out/r05z_window.txt
for case in poly4 poly12 poly20 poly64; do
  for v in default= w256=SORT_LEVELS:1,SORT_WINDOW:256 w1k=SORT_LEVELS:1,SORT_WINDOW:1024 w4k=SORT_LEVELS:1,SORT_WINDOW:4096 whole=SORT_LEVELS:1; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_window.txt 2>&1
  done
done
for case in ragged4 ragged32; do
  for v in default= w256=SORT_LEVELS:1,SORT_WINDOW:256 w1k=SORT_LEVELS:1,SORT_WINDOW:1024 w4k=SORT_LEVELS:1,SORT_WINDOW:4096 default=; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_window.txt 2>&1
  done
done
cat gpurun_out/r05z_window.txt
