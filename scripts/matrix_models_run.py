#!/usr/bin/env python3
"""Marginal and joint pass times of the materialised-P kernels (HKY, eigen models outside 16 <= k <= 32) at size."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import pastml_oracle as orc  # noqa: E402  (only its diagonalisation helper, to build a model)
from pastml_amd import hip, synthetic  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 18
for kind, k, C in (('HKY', 4, 8), ('EIGEN', 5, 8), ('EIGEN', 12, 8), ('EIGEN', 40, 2), ('F81', 4, 8), ('F81', 12, 8)):
    rng = np.random.default_rng(k)
    flat = synthetic.balanced_forest(levels)
    pi = rng.dirichlet(np.ones(k) * 3)
    if kind == 'F81':
        spec = dict(kind=0, pi=pi)
    elif kind == 'HKY':
        spec = dict(kind=1, pi=pi, kappa=3.0)
    else:
        R = np.triu(rng.uniform(0.05, 3, size=(k, k)), 1)
        d, a, ainv = orc.diagonalise(pi, R + R.T)
        spec = dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv)
    eng = hip.Engine(flat, C, k)
    eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))

    def marg():
        eng.set_models([(spec, (1.0, 0.0, 1.0))] * C)
        eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
        eng.sync()

    def joint():
        eng.set_models([(spec, (1.0, 0.0, 1.0))] * C)
        eng.bottom_up(False)
        eng.joint_backtrace(copy_out=False)
        eng.sync()
    out = []
    for fn in (marg, joint):
        fn()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        out.append((time.perf_counter() - t0) / 3 * 1e3)
    print('%-5s k %2d C %d tips %d: marginal %.3f ms, joint %.3f ms' % (kind, k, C, flat.n_tips, out[0], out[1]))
    eng.close()
