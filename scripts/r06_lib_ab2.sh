#!/bin/bash
# like r06_lib_ab.sh with environment switches per variant: scripts/r06_lib_ab2.sh <out> "<lib> <ENV=..>" ...
out=$1; shift
: > $out
for rep in 1 2; do
  for spec in "$@"; do
    set -- $spec; lib=$1; shift
    [ "$lib" = "base" ] && L="" || L=$PWD/scratch/r06/lib_$lib.so
    env "$@" PASTML_HIP_LIBRARY=$L timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-validate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['roofline_bottom_up']; t=d['roofline_top_down']
print('%-28s step %.3f ms  BU two-level %.3f  BU levels %.3f  TD two-level %.3f  TD levels %.3f  prep %.3f  lnL %.9f' % ('$spec', d['ms_per_step'], b['two_level_ms_per_step'], b['level_kernel_ms_per_step'], t['two_level_ms_per_step'], t['level_kernel_ms_per_step'], d['kernel_ms_per_step']['prep'], d['loglik_sum']))
" >> $out
  done
done
cat $out
