"""Fused (one-matrix) against materialised sweeps of 65 - 128-state eigen models on a small ragged tree, by number of columns."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest
from pastml_amd.models._eigen import get_diagonalisation
n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 3619
flat = FlatForest.random(n_tips, seed=5, max_arity=3)
print('random tree, {} tips, {} nodes; ms per bottom-up sweep incl. the model upload'.format(flat.n_tips, flat.n_nodes))
for k in (67, 128):
    rng = np.random.default_rng(k)
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    for C in (1, 8, 64):
        specs = []
        for c in range(C):
            pi = rng.dirichlet(np.ones(k) * 4)
            d, a, ainv = get_diagonalisation(pi, rates)
            specs.append((dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv), (1.0, 0.0, 1.0)))
        tips = np.stack([rng.integers(0, k, size=flat.n_tips).astype(np.int32) for c in range(C)])
        out = {}
        for name, tune in (('hbm', dict(NO_EIGEN_GEMM=1)), ('fused', {})):
            with hip.Engine(flat, C, k, tune=tune) as eng:
                eng.set_tip_states(tips)
                def bu():
                    eng.set_models(specs)
                    return eng.bottom_up(True)
                bu(); bu(); eng.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    r = bu()
                eng.sync()
                out[name] = (time.perf_counter() - t0) / 5 * 1e3
        print('k = {:3d}  {:2d} columns   P(t) in HBM {:8.2f} ms   fused {:8.2f} ms   x {:.2f}'.format(k, C, out['hbm'], out['fused'], out['hbm'] / out['fused']), flush=True)
