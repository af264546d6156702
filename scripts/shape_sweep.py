#!/usr/bin/env python3
"""Marginal pass over a few shapes (balanced tree, F81): ms per pass and node*state*char/s, to spot pathological shapes."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

for levels, k, C in ((18, 2, 32), (18, 4, 32), (18, 5, 32), (18, 12, 32), (18, 20, 32), (18, 33, 32), (18, 64, 32),
                     (18, 100, 8), (16, 200, 8), (20, 4, 64), (20, 20, 32)):
    flat = synthetic.balanced_forest(levels)
    eng = hip.Engine(flat, C, k)
    specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
    eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))

    def step():
        eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
        eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
        eng.sync()
    for _ in range(2):
        step()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    units = flat.n_nodes * k * C
    gb = units * 48 / 1e9
    print('levels %2d k %3d C %2d: %7.3f ms  %.3g units/s  algorithmic %.0f GB/s' % (levels, k, C, ms, units / ms * 1e3, gb / ms * 1e3))
    eng.close()
