#!/bin/bash
set -e
: > gpurun_out/r05u_thin_rest.txt
for case in balanced4 balanced12 poly64 ragged20 cfg2; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $case default= off=NO_THIN:1 narrow=NO_THIN_WIDE:1 >> gpurun_out/r05u_thin_rest.txt 2>&1
done
cat gpurun_out/r05u_thin_rest.txt
