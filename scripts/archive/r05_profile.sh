#!/bin/bash
# round 5 profiles: cfg4 kernel trace + HBM counters (profile_cfg4.sh), the P(t) batch's kernel trace and counters, the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
cd $R || exit 1
mkdir -p gpurun_out
bash scripts/profile_cfg4.sh cfg4 > gpurun_out/${TAG}_profile_cfg4.log 2>&1
echo "profile_cfg4 rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pij_kt $R/gpurun_out/pij_fetch $R/gpurun_out/pij_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pij_kt -o run -- python3 $R/scripts/r05_pij.py 20 32 > $R/gpurun_out/${TAG}_pij_kt.txt 2>&1
echo "pij kt rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pij_fetch -o run -- python3 $R/scripts/r05_pij.py 20 32 > $R/gpurun_out/${TAG}_pij_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pij_write -o run -- python3 $R/scripts/r05_pij.py 20 32 > $R/gpurun_out/${TAG}_pij_write.txt 2>&1
echo "pij pmc rc=$?"
cd $R
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_full.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['speedup_vs_cpu_baseline'])
s=d['secondary']
print(s.get('error'))
print('ragged', s['ragged262k']['k64']['ms_per_pass'], s['ragged262k']['k4']['ms_per_pass'], 'cfg3', s['cfg3']['ms_joint_sweep'], 'cfg2', s['cfg2']['ms_per_pass'], 'grad', s['cfg5_gradient']['ms_per_gradient'])
print('pij', {k:(v['ms_per_batch'], v['roofline']['frac']) for k,v in s['pij_batch'].items() if isinstance(v,dict)})
print('cfg5', s['cfg5_acr']['seconds'], s['cfg5_acr']['columns_beyond_1e6'], s['cfg5_acr']['worst_rel_loglik_shortfall'])
PY
