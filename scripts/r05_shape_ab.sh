#!/bin/bash
: > gpurun_out/r05z_cfg4_libab.txt
for rep in 1 2; do
  for lib in "" "$PWD/scratch/r05w/libpastml_hip_oldseq.so"; do
    PASTML_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=%s' % ('$lib'[-14:] or 'new'), d['ms_per_step'], d['roofline']['frac'], d['library']['build_digest'])" >> gpurun_out/r05z_cfg4_libab.txt
  done
done
cat gpurun_out/r05z_cfg4_libab.txt
