#!/bin/bash
# A/B of two builds of the library inside one call: pastml_amd/libpastml_hip.so (A) against scratch/$1 (B);
# $2... = what to run per variant (default: the ragged / balanced 262 144-tip passes and a short cfg4 bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
B=$R/scratch/$1
shift
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$B;; esac
  echo "== $v"
  timeout -k 10 300 python3 $R/scripts/r04_ragged.py "$@" 2> $O/r04ab_$v.err || { cat $O/r04ab_$v.err; exit 1; }
  if [ -z "$NO_BENCH" ]; then
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/libab_$v.json 2> $O/libab_$v.err || { tail -5 $O/libab_$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/libab_$v.json')); print('$v cfg4', round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'bu frac', round(d['roofline_bottom_up']['frac'],4), 'td frac', round(d['roofline']['frac'],4))"
  fi
done
unset PASTML_HIP_LIBRARY
