#!/bin/bash
# SQ counters of the LARGEST eigen_fused launch of a cfg3 joint sweep (the 131 072-node level), three PMC passes + a
# kernel trace.  Output: gpurun_out/$1_{kt,a,b,c}; summarise with scripts/pmc_cfg3_big.py gpurun_out/$1
T=${1:-big}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_kt $R/gpurun_out/${T}_a $R/gpurun_out/${T}_b $R/gpurun_out/${T}_c
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_kt -o run -- python3 $R/scripts/cfg3_run.py ${MODE:-j} 3 > $R/gpurun_out/${T}_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/${T}_a -o run -- python3 $R/scripts/cfg3_run.py ${MODE:-j} 3 > $R/gpurun_out/${T}_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/${T}_b -o run -- python3 $R/scripts/cfg3_run.py ${MODE:-j} 3 > $R/gpurun_out/${T}_b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/${T}_c -o run -- python3 $R/scripts/cfg3_run.py ${MODE:-j} 3 > $R/gpurun_out/${T}_c.log 2>&1
for f in kt a b c; do tail -n 1 $R/gpurun_out/${T}_$f.log; done
