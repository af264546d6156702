"""P(t) batch and marginal pass of eigen models beyond 32 states: pij_eigen_wide_kernel against the kernel it replaces (NO_PIJ_WIDE).
balanced 16 384-tip tree x 4 characters; python scripts/r06_pij_wide.py [k ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastml_amd import hip, synthetic
from pastml_amd.models._eigen import get_diagonalisation

ks = [int(a) for a in sys.argv[1:]] or [36, 48, 64, 67, 100, 128, 200, 256]
flat = synthetic.balanced_forest(14)
C = 4
print('balanced {}-tip tree ({} branches) x {} characters; ms per call, wall clock around the C-ABI'.format(flat.n_tips, flat.n_nodes, C))
print('{:>4s} {:>22s} {:>22s} {:>10s} {:>12s} {:>22s} {:>22s}'.format('k', 'P(t) batch, generic', 'P(t) batch, wide', 'x', 'TFLOP/s', 'marginal pass, generic', 'marginal pass, wide'))
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
    for name, tune in (('generic', dict(NO_PIJ_WIDE=1, NO_EIGEN_GEMM=1, NO_EIGEN_JOINT_VALU=1)), ('wide', dict(NO_EIGEN_GEMM=1, NO_EIGEN_JOINT_VALU=1))):
        with hip.Engine(flat, C, k, tune=tune) as eng:
            eng.set_tip_states(tips)
            eng.set_models(specs)
            def pij():
                eng.set_models(specs)
                eng.pij_batch(copy_out=False)
            def marginal():
                eng.set_models(specs)
                return eng.marginal_pass(posterior=False, lh=False)[0]
            res = []
            for fn in (pij, marginal):
                reps = 2 if (name == 'generic' and k > 64) else 5
                fn(); eng.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    r = fn()
                eng.sync()
                res.append((time.perf_counter() - t0) / reps * 1e3)
            out[name] = res + [r]
    flops = 2.0 * k ** 3 * flat.n_nodes * C
    rel = float(np.max(np.abs((out['wide'][2] - out['generic'][2]) / out['generic'][2])))
    print('{:4d} {:22.2f} {:22.2f} {:10.1f} {:12.1f} {:22.2f} {:22.2f}   ln L rel. diff {:.1e}'.format(
        k, out['generic'][0], out['wide'][0], out['generic'][0] / out['wide'][0], flops / (out['wide'][0] * 1e-3) / 1e12,
        out['generic'][1], out['wide'][1], rel), flush=True)
