#!/bin/bash
# grid cap of eigen_gemm_kernel: scripts/r06_eig_blocks.sh <out>
out=$1; : > $out
for a in "16 61 32" "16 61 4" "16 36 16" "16 20 32" "18 20 1" "16 8 32"; do
  for cap in 16384 4096 2048 1024 512; do
    PASTML_HIP_EIG_BLOCKS=$cap python scripts/r06_eigen_k61.py $a 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s cap %5d  marginal %.3f ms  bottom-up %.3f ms  hbm frac %.3f' % ('$a', $cap, d['ms_marginal_pass'], d['ms_bottom_up_sweep'], r['frac']))
" >> $out
  done
done
cat $out
