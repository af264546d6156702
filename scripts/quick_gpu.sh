#!/bin/bash
# quick GPU check used while iterating: parity tests, one bench line, a kernel trace of two steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/pytest_gpu.log 2>&1; tail -3 $R/gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_quick.json 2> $R/gpurun_out/bench_quick.err
rm -rf $R/gpurun_out/kt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/kt.err
python3 $R/scripts/kt_levels.py $R/gpurun_out/kt
python3 -c "
import json; d=json.load(open('$R/gpurun_out/bench_quick.json')); print(d['ms_per_step'], d['kernel_ms_per_step'])"
