#!/bin/bash
# Kernel stats and SQ counters of the kernels outside the headline: the two-GEMM marginal sweeps of the eigen models
# (eigen_gemm_kernel: the MFMA kernel of the default dispatch), the joint sweep (eigen_joint_kernel), the small-k F81
# level kernels, the thin-level kernels (cfg2, cfg5-shaped gradient).  Outputs under gpurun_out/r03e_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counters or "", program args...
  tag=$1; pmc=$2; shift 2
  rm -rf $O/r03e_$tag
  if [ -z "$pmc" ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03e_$tag -o run -- python3 "$@" > $O/r03e_$tag.log 2>&1 || exit 1
  else
    timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/r03e_$tag -o run -- python3 "$@" > $O/r03e_$tag.log 2>&1 || exit 1
  fi
}
run cfg3m_kt "" $R/scripts/cfg3_run.py m 10
run cfg3m_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" $R/scripts/cfg3_run.py m 3
run cfg3m_sq_b "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" $R/scripts/cfg3_run.py m 3
run cfg3j_kt "" $R/scripts/cfg3_run.py j 10
run cfg3j_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" $R/scripts/cfg3_run.py j 3
run smallk_kt "" $R/scripts/k4_run.py 18 4 32
run smallk_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM" $R/scripts/k4_run.py 18 4 32
run thin_kt "" $R/scripts/r03_thin_trace.py both 20
run thin_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" $R/scripts/r03_thin_trace.py both 5
python3 $R/scripts/r03_summarise_secondary.py $O > $O/r03e_secondary_counters.md
tail -60 $O/r03e_secondary_counters.md
