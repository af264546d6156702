#!/bin/bash
: > gpurun_out/r05z_binsize.txt
for case in poly3_20 poly3_64 poly20 ragged20 mid64 hiv67c204 hiv30c93; do
  for v in default= s32=THIN_BLOCK_NODES:32 s64=THIN_BLOCK_NODES:64 s128=THIN_BLOCK_NODES:128 s512=THIN_BLOCK_NODES:512; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_binsize.txt 2>&1
  done
done
cat gpurun_out/r05z_binsize.txt
