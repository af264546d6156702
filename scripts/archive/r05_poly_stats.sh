#!/bin/bash
# rocprofv3 kernel stats of the polytomy forest's marginal pass (secondary.polytomies100k): which kernels in which lane shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for case in poly3_64 poly3_20; do
  rm -rf $O/r05poly_$case
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05poly_$case -o run -- python3 $R/scripts/r04_prof_driver.py $case 20 > $O/r05poly_$case.log 2>&1 || { tail -5 $O/r05poly_$case.log; exit 1; }
  find $O/r05poly_$case -name '*kernel_stats.csv' -exec cp {} $O/r05poly_${case}_kernel_stats.csv \;
  find $O/r05poly_$case -name '*.csv' -size +2M -delete
  head -12 $O/r05poly_${case}_kernel_stats.csv | cut -c1-150
done
