#!/bin/bash
# round 5, first GPU call: the whole GPU suite, the full bench line, and the split-pass A/B (parts = 1 / 2 / 4)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r05a_pytest.log
tail -5 gpurun_out/r05a_pytest.log
for parts in 1 2 4; do
  PASTML_HIP_SPLIT_PARTS=$parts timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline \
      > gpurun_out/r05a_split_parts$parts.json 2> gpurun_out/r05a_split_parts$parts.err
  echo "parts=$parts rc=$?"
  python - <<PY
import json
d=json.load(open('gpurun_out/r05a_split_parts$parts.json'))
print('parts $parts', d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'], d['validation'].get('against_reference_run'))
PY
done
timeout -k 10 600 python bench.py > gpurun_out/r05a_bench_full.json 2> gpurun_out/r05a_bench_full.err
echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r05a_bench_full.json'))
print(d['ms_per_step'], d['roofline']['frac'])
s=d['secondary']
print({k:(v.get('ms_per_pass') or v.get('seconds') or v.get('ms_per_gradient') or v.get('ms_per_step')) for k,v in s.items() if isinstance(v,dict)})
print(s.get('pij_batch')); print(s.get('error'))
print(s['ragged262k']['k64']['ms_per_pass'], s['ragged262k']['k4']['ms_per_pass'], s['cfg3']['ms_joint_sweep'])
print(s['cfg5_acr']['seconds'], s['cfg5_acr']['columns_beyond_1e6'], s['cfg5_acr']['worst_rel_loglik_shortfall'])
PY
