#!/bin/bash
# SQ counters of one bench step (two PMC passes); output: gpurun_out/pmc_sq_{a,b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq_a $R/gpurun_out/pmc_sq_b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_a -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null 2> $R/gpurun_out/pmc_sq_a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_b -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null 2> $R/gpurun_out/pmc_sq_b.err
ls $R/gpurun_out/pmc_sq_a $R/gpurun_out/pmc_sq_b
