#!/bin/bash
# cfg3 correctness (the eigen tests) + timing, one call
mkdir -p gpurun_out/q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "cfg3 or jtt or EIGEN or eigen or determinism or random_forests" > gpurun_out/q/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/q/pytest.log
python scripts/cfg3_run.py j 5 2>&1 | tail -3
python - <<'PY'
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
def timed(fn, reps=30):
    fn(); eng.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    eng.sync(); return (time.perf_counter() - t0) / reps * 1e3
print('graph replay: joint sweep %.3f ms, + backtrace %.3f ms, marginal BU %.3f ms, BU+TD %.3f ms' % (
    timed(lambda: eng.bottom_up(False)), timed(lambda: (eng.bottom_up(False), eng.joint_backtrace(copy_out=False))),
    timed(lambda: eng.bottom_up(True)), timed(lambda: (eng.bottom_up(True), eng.top_down_marginals(posterior=False, lh=False)))))
PY
