#!/usr/bin/env python3
"""cfg4-size tree, k=64, few characters: the device sequence of ml_acr for MPPA with force_joint (marginal pass, joint
sweep + back-trace, MPPA selection, restricted-likelihood sweep), for a kernel trace."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = 64
flat = synthetic.balanced_forest(levels)
eng = hip.Engine(flat, C, k)
specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
eng.profile_enable(True)
for rep in range(2):
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    eng.bottom_up(False)
    eng.joint_backtrace(copy_out=False)
    eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    sel, nsel = eng.select_states('MPPA', force_joint=True)
    lnl = eng.bottom_up(True)
    print('restricted lnL', lnl[:2], 'avg states per node', nsel.mean())
eng.close()
