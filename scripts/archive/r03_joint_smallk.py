#!/usr/bin/env python3
"""Joint against marginal bottom-up sweep at small k (argv: levels k C): where the joint sweep stands."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

levels, k, C = (int(x) for x in sys.argv[1:4])
flat = synthetic.balanced_forest(levels)
with hip.Engine(flat, C, k) as eng:
    specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
    eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
    eng.set_models(specs)

    def timed(fn, reps=10):
        fn(); eng.sync(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        return (time.perf_counter() - t0) / reps * 1e3
    print('k', k, 'marginal BU %.3f ms, joint BU %.3f ms, joint pass %.3f ms, marginal pass %.3f ms' % (
        timed(lambda: eng.bottom_up(True)), timed(lambda: eng.bottom_up(False)),
        timed(lambda: eng.joint_pass(copy_out=False)), timed(lambda: eng.marginal_pass(posterior=False, lh=False))))
