#!/bin/bash
# round 3 A/B: GPU tests, then the cfg4 bench with and without the folded per-branch pass, then a kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest $R/tests -m gpu -x -q > $O/r03_pytest.log 2>&1; rc=$?; tail -5 $O/r03_pytest.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for v in nofold fold nofold2; do
  if [ $v = fold ]; then export PASTML_HIP_FOLD=1; else unset PASTML_HIP_FOLD; fi
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r03_bench_$v.json 2> $O/r03_bench_$v.err || exit 1
  python3 -c "
import json; d=json.load(open('$O/r03_bench_$v.json')); print('$v', round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'bu frac', d['roofline_bottom_up']['frac'], 'td frac', d['roofline']['frac'])"
done
unset PASTML_HIP_FOLD
rm -rf $O/r03_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_kt -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/r03_kt.err
python3 $R/scripts/kt_levels.py $O/r03_kt | tail -40
export PASTML_HIP_FOLD=1
rm -rf $O/r03_kt_fold
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_kt_fold -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/r03_kt_fold.err
python3 $R/scripts/kt_levels.py $O/r03_kt_fold | tail -40
