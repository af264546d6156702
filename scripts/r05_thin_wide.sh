#!/bin/bash
set -e
: > gpurun_out/r05u_thin_wide.txt
for case in ragged64 poly64 ragged20 poly20 mid64; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $case default= wide=THIN_WIDE:1 wide512=THIN_WIDE:1,THIN_BLOCK_NODES:512 wide128=THIN_WIDE:1,THIN_BLOCK_NODES:128 >> gpurun_out/r05u_thin_wide.txt 2>&1
done
cat gpurun_out/r05u_thin_wide.txt
