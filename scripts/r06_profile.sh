#!/bin/bash
# round 6 profiles: cfg4 kernel trace + HBM counters (profile_cfg4.sh), the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
cd $R || exit 1
mkdir -p gpurun_out
bash scripts/profile_cfg4.sh cfg4 > gpurun_out/${TAG}_profile_cfg4.log 2>&1
echo "profile_cfg4 rc=$?"
cd $R
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_full.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['speedup_vs_cpu_baseline'])
PY
