#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ...   -- one bench line per value of an environment variable
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/sweep_$VAR_$v.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$R/gpurun_out/sweep_$VAR_$v.json')); print('$VAR=$v', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
