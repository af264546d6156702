#!/usr/bin/env python3
"""cfg2 (65 536 tips, JC k=4, 1 character): full marginal pass latency, graph replay on (the product default)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

flat = synthetic.balanced_forest(16)
eng = hip.Engine(flat, 1, 4)
spec = dict(kind=0, pi=np.ones(4) / 4)
eng.set_models([(spec, (1.0, 0.0, 1.0))])
eng.set_tip_states(synthetic.tip_states(flat.n_tips, 4, 0))


def step():
    eng.set_models([(spec, (1.0, 0.0, 1.0))])
    lnl = eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    eng.sync()
    return lnl


for _ in range(5):
    lnl = step()
t0 = time.perf_counter()
for _ in range(200):
    step()
print('cfg2 marginal ms', (time.perf_counter() - t0) / 200 * 1e3, 'lnL', lnl[0], 'narrow', os.environ.get('PASTML_HIP_NARROW_UNITS'))
eng.close()
