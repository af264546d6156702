#!/bin/bash
: > gpurun_out/r05z_hiv.txt
for case in hiv12 hiv2; do
  for v in default= b128=BLOCK_NODES:128 b512=BLOCK_NODES:512 b1k=BLOCK_NODES:1024 t128=BLOCK_THREADS:128 t256=BLOCK_THREADS:256 nocap=BLOCK_HEIGHT_CAP:0 thin=BLOCK_NODES:0,THIN_UNITS:4096 default=; do
    timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05z_hiv.txt 2>&1
  done
done
cat gpurun_out/r05z_hiv.txt
