#!/bin/bash
# same-box A/B of library builds: $1 = cases, the libraries: the one in place, then scratch/r05w/*.so
: > gpurun_out/r05z_libab3.txt
for case in $1; do
  for rep in 1 2; do
    for lib in "" $PWD/scratch/r05w/libpastml_hip_prev.so $PWD/scratch/r05w/libpastml_hip_oldseq.so; do
      echo -n "lib=${lib##*/} " >> gpurun_out/r05z_libab3.txt
      PASTML_HIP_LIBRARY=$lib timeout -k 10 120 python scripts/r05_tune_one.py $case default= >> gpurun_out/r05z_libab3.txt 2>&1
    done
  done
done
cut -c1-150 gpurun_out/r05z_libab3.txt
