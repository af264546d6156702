#!/bin/bash
set -e
: > gpurun_out/r05u_thin_super.txt
for case in ragged64 poly64; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $case default= nosuper=NO_SUPER:1 nostack=NO_STACK:1 nosuper_nothin=NO_SUPER:1,NO_THIN:1 >> gpurun_out/r05u_thin_super.txt 2>&1
done
cat gpurun_out/r05u_thin_super.txt
