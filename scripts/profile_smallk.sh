#!/bin/bash
# HBM counters + kernel trace for the small-k many-column case (args: levels k C)
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-18}; K=${2:-4}; C=${3:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sk_fetch $R/gpurun_out/sk_write $R/gpurun_out/sk_sq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/sk_fetch -o run -- python3 $R/scripts/k4_run.py $L $K $C > $R/gpurun_out/sk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/sk_write -o run -- python3 $R/scripts/k4_run.py $L $K $C > $R/gpurun_out/sk_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/sk_sq -o run -- python3 $R/scripts/k4_run.py $L $K $C > $R/gpurun_out/sk_sq.log 2>&1
tail -2 $R/gpurun_out/sk_sq.log
