#!/bin/bash
: > gpurun_out/r05z_thin_g16.txt
for case in poly3_20 poly3_64 poly20 poly64; do
  for v in default= off=NO_THIN:1 t2k=THIN_UNITS:2048 t1k=THIN_UNITS:1024 t512=THIN_UNITS:512; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_thin_g16.txt 2>&1
  done
done
cat gpurun_out/r05z_thin_g16.txt
