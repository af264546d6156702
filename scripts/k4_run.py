#!/usr/bin/env python3
"""Balanced tree, small k: marginal pass for a kernel trace (argv: levels k C)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

levels, k, C = (int(x) for x in sys.argv[1:4])
flat = synthetic.balanced_forest(levels)
eng = hip.Engine(flat, C, k)
specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
eng.profile_enable(True)
for _ in range(3):
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    eng.sync()
eng.close()
