#!/bin/bash
# cfg3 joint sweep: correctness of the eigen paths, timing, per-kernel trace (one call)
mkdir -p gpurun_out/q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "cfg3 or jtt or EIGEN or eigen or determinism or random_forests" > gpurun_out/q/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/q/pytest.log
cat > /tmp/t.py <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX
from pastml_amd.models.generator import get_diagonalisation
flat = synthetic.balanced_forest(18)
d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
eng = hip.Engine(flat, 1, 20)
eng.set_models([(spec, (1.0, 0.0, 1.0))])
eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
def timed(fn, reps=50):
    fn(); eng.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.sync(); return (time.perf_counter() - t0) / reps * 1e3
print(os.environ.get('TAG', ''), 'joint sweep %.4f ms, + backtrace %.4f ms lnL %.10f' % (
    timed(lambda: eng.bottom_up(False)), timed(lambda: (eng.bottom_up(False), eng.joint_backtrace(copy_out=False))), eng.bottom_up(False)[0]))
PY
for i in 1 2; do
TAG=new python /tmp/t.py
TAG=old PASTML_HIP_NO_EIGJ_TIERS=1 python /tmp/t.py
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q/kt -o run -- python3 $GRAFT_REPO_ROOT/scripts/cfg3_run.py j 10 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/q/kt/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:10]:
    print('%-60s calls %4s avg %8.1f us min %8.1f max %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
