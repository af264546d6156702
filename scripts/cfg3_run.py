#!/usr/bin/env python3
"""cfg3 (262 144 tips, JTT k=20): joint (mode j) or marginal (mode m) passes, for profiling."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX  # noqa: E402
from pastml_amd.models.generator import get_diagonalisation  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'j'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 18
flat = synthetic.balanced_forest(levels)
d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
eng = hip.Engine(flat, 1, 20)
eng.set_models([(spec, (1.0, 0.0, 1.0))])
eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
eng.profile_enable(True)  # direct submission (no graph replay): every launch shows in the trace
for i in range(reps + 1):
    t0 = time.perf_counter()
    eng.set_models([(spec, (1.0, 0.0, 1.0))])
    if mode == 'j':
        lnl = eng.bottom_up(False)
        eng.joint_backtrace(copy_out=False)
    else:
        lnl = eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
    eng.sync()
    print(mode, 'ms', (time.perf_counter() - t0) * 1e3, 'lnL', lnl[0])
eng.close()
