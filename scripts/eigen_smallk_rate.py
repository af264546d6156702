#!/usr/bin/env python3
"""Eigen model (random reversible rate matrix), balanced tree: marginal pass time (argv: levels k C)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.models.generator import get_diagonalisation  # noqa: E402

levels, k, C = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(1)
flat = synthetic.balanced_forest(levels)
specs = []
for c in range(C):
    pi = rng.dirichlet(np.ones(k) * 3)
    R = np.triu(rng.uniform(0.1, 3, size=(k, k)), 1)
    d, A, Ainv = get_diagonalisation(pi, R + R.T)
    specs.append((dict(kind=2, pi=pi, d=d, A=A, Ainv=Ainv), (1.0, 0.0, 1.0)))
eng = hip.Engine(flat, C, k)
eng.set_models(specs)
eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))


def timed(fn, reps=10):
    fn(); eng.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3


print('k=%d C=%d tips=%d gemm=%s: bottom-up %.3f ms, marginal pass %.3f ms, joint pass %.3f ms' % (
    k, C, flat.n_tips, 'off' if os.environ.get('PASTML_HIP_NO_EIGEN_GEMM') else 'on',
    timed(lambda: eng.bottom_up(True)), timed(lambda: eng.marginal_pass(posterior=False, lh=False)),
    timed(lambda: eng.joint_pass(copy_out=False))))
eng.close()
