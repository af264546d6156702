"""Sum sweeps of eigen models with 65 - 128 states: the one-matrix two-GEMM kernels against P(t) materialised in HBM (built by the
matrix-core batch of r06u).  balanced 16 384-tip tree x 4 characters; python scripts/r06_eigen_fused_wide.py [k ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastml_amd import hip, synthetic
from pastml_amd.models._eigen import get_diagonalisation

ks = [int(a) for a in sys.argv[1:]] or [67, 80, 96, 100, 112, 128]
levels = int(os.environ.get("LEVELS", "14"))
flat = synthetic.balanced_forest(levels)
C = 4
print('balanced {}-tip tree ({} nodes) x {} characters; ms per call incl. the model upload, wall clock around the C-ABI'.format(flat.n_tips, flat.n_nodes, C))
print('{:>4s} {:>28s} {:>28s} {:>8s} {:>28s} {:>28s} {:>8s}'.format('k', 'bottom-up, P(t) in HBM', 'bottom-up, fused', 'x', 'marginal pass, P(t) in HBM', 'marginal pass, fused', 'x'))
for k in ks:
    rng = np.random.default_rng(k)
    rates = np.triu(rng.uniform(0.05, 3.0, size=(k, k)), 1)
    rates = rates + rates.T
    specs = []
    for c in range(C):
        pi = rng.dirichlet(np.ones(k) * 4)
        d, a, ainv = get_diagonalisation(pi, rates)
        specs.append((dict(kind=2, pi=pi, d=d, A=a, Ainv=ainv), (1.0, 0.0, 1.0)))
    tips = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
    out = {}
    for name, tune in (('hbm', dict(NO_EIGEN_GEMM=1)), ('fused', {})):
        with hip.Engine(flat, C, k, tune=tune) as eng:
            eng.set_tip_states(tips)
            def bu():
                eng.set_models(specs)
                return eng.bottom_up(True)
            def marginal():
                eng.set_models(specs)
                return eng.marginal_pass(posterior=False, lh=False)[0]
            res = []
            for fn in (bu, marginal):
                fn(); eng.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    r = fn()
                eng.sync()
                res.append((time.perf_counter() - t0) / 5 * 1e3)
            out[name] = res + [r]
    rel = float(np.max(np.abs((out['fused'][2] - out['hbm'][2]) / out['hbm'][2])))
    print('{:4d} {:28.2f} {:28.2f} {:8.1f} {:28.2f} {:28.2f} {:8.1f}   ln L rel. diff {:.1e}'.format(
        k, out['hbm'][0], out['fused'][0], out['hbm'][0] / out['fused'][0], out['hbm'][1], out['fused'][1],
        out['hbm'][1] / out['fused'][1], rel), flush=True)
