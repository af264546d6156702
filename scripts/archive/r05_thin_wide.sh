#!/bin/bash
set -e
: > gpurun_out/r05u_thin_bytes.txt
for case in ${CASES:-ragged64 poly64 ragged20 ragged4}; do
  for rep in 1 2; do
    for v in off=NO_THIN:1 t512=THIN_UNITS:512 t1k=THIN_UNITS:1024 t1500=THIN_UNITS:1500 t2k=THIN_UNITS:2048 t3k=THIN_UNITS:3000 t4k=THIN_UNITS:4096; do
      timeout -k 10 120 python scripts/r05_tune_one.py $case $v >> gpurun_out/r05u_thin_bytes.txt 2>&1
    done
  done
done
cat gpurun_out/r05u_thin_bytes.txt
