#!/bin/bash
# Profiles BASELINE config 3 (262 144 tips, JTT k=20, joint sweep on the fused FP64 matrix-core kernels) on the GPU box:
# kernel trace + stats, then SQ counters (separate pass).  Outputs: gpurun_out/cfg3_kt, gpurun_out/cfg3_pmc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cfg3_kt $R/gpurun_out/cfg3_pmc $R/gpurun_out/cfg3_pmc2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg3_kt -o run -- python3 $R/scripts/cfg3_run.py j 10 > $R/gpurun_out/cfg3_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/cfg3_pmc -o run -- python3 $R/scripts/cfg3_run.py j 3 > $R/gpurun_out/cfg3_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/cfg3_pmc2 -o run -- python3 $R/scripts/cfg3_run.py j 3 > $R/gpurun_out/cfg3_pmc2.log 2>&1
tail -3 $R/gpurun_out/cfg3_kt.log; tail -3 $R/gpurun_out/cfg3_pmc.log; tail -3 $R/gpurun_out/cfg3_pmc2.log
