#!/bin/bash
: > gpurun_out/r05z_poly2.txt
for case in bin100k_4 poly3_4 poly4 bin100k_12 poly3_12 poly12 ragged4 ragged12 hiv12 cfg2; do
  timeout -k 10 120 python scripts/r05_tune_one.py $case default= >> gpurun_out/r05z_poly2.txt 2>&1
done
cat gpurun_out/r05z_poly2.txt
