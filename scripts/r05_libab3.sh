#!/bin/bash
: > gpurun_out/r05z_libab2.txt
for case in cfg2 hiv12 hiv40 hiv2 midpoly3_4 ragged4; do
  for rep in 1 2; do
    for lib in "" "$PWD/scratch/r05w/libpastml_hip_prev.so" "$PWD/scratch/r05w/libpastml_hip_oldseq.so"; do
      echo -n "lib=${lib##*/} " >> gpurun_out/r05z_libab2.txt
      PASTML_HIP_LIBRARY=$lib timeout -k 10 120 python scripts/r05_tune_one.py $case default= >> gpurun_out/r05z_libab2.txt 2>&1
    done
  done
done
cat gpurun_out/r05z_libab2.txt
