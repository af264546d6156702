#!/bin/bash
# A/B of two builds of the library inside one call: pastml_amd/libpastml_hip.so (A) against scratch/$1 (B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
B=$R/scratch/$1
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$B;; esac
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/libab_$v.json 2> $O/libab_$v.err || exit 1
  python3 -c "
import json; d=json.load(open('$O/libab_$v.json')); print('$v', round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'bu frac', round(d['roofline_bottom_up']['frac'],4), 'td frac', round(d['roofline']['frac'],4))"
done
unset PASTML_HIP_LIBRARY
