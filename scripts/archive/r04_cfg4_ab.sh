#!/bin/bash
# cfg4 step of the in-tree library (A) against scratch/$1 (B), alternating; kernel times by HIP events from bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/cfg4ab_$v.json 2> $O/cfg4ab_$v.err || { tail -5 $O/cfg4ab_$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/cfg4ab_$v.json')); print('$v cfg4', round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'bu frac', round(d['roofline_bottom_up']['frac'],4), 'td frac', round(d['roofline']['frac'],4), 'validation', d['validation']['against_reference_run'] if 'validation' in d else '')"
done
unset PASTML_HIP_LIBRARY
