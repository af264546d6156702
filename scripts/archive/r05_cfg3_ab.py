"""cfg3 (and two ragged trees) joint sweep: engines with different switches in one process, timed alternately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX
from pastml_amd.models.generator import get_diagonalisation

d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
cases = dict(cfg3=(lambda: synthetic.balanced_forest(18), 1), ragged=(lambda: FlatForest.random(40000, seed=4, max_arity=2), 1),
             ragged8=(lambda: FlatForest.random(100000, seed=4, max_arity=3), 8))
variants = [('pipe', {}), ('plain', dict(NO_EIGJ_PIPE=1)), ('pipe-again', {})]
for name in sys.argv[1:] or ['cfg3', 'ragged', 'ragged8']:
    make, C = cases[name]
    f = make()
    engines = []
    for vname, tune in [('(ballast)', {})] + variants:
        eng = hip.Engine(f, C, 20, tune=tune)
        eng.set_models([(spec, (1.0, 0.0, 1.0))] * C)
        eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, 20, c) for c in range(C)]))
        for _ in range(3):
            lnl, js = eng.joint_pass()
        engines.append((vname, eng, lnl, js, []))
    for rnd in range(3):
        for vname, eng, lnl, js, times in engines:
            eng.sync(); t0 = time.perf_counter()
            for _ in range(50):
                eng.bottom_up(False)
            eng.sync(); times.append((time.perf_counter() - t0) / 50 * 1e3)
    ref = engines[1]
    for vname, eng, lnl, js, times in engines[1:]:
        print('%-8s %-11s joint sweep %s ms (min %.4f)  same ln L %s, same states %s'
              % (name, vname, ' '.join('%.4f' % v for v in times), min(times), np.array_equal(lnl, ref[2]), np.array_equal(js, ref[3])), flush=True)
    for e in engines:
        e[1].close()
