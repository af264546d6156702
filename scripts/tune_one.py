"""
One schedule variant per PROCESS (ballast engine first, then the timed one: every variant's buffers land where the others' do --
engines created one after the other in a process differ by up to 6 % with identical settings).
usage: tune_one.py <case> <name=SWITCH:value,...>        cases: scripts/tune_ab.py
"""
import os, sys, time, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest

src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tune_ab.py')).read()
ns = dict(globals())
exec(src[src.index('def hiv1c_forest'):src.index('make, k, C = cases')], ns)
make, k, C = ns['cases'][sys.argv[1]]
name, _, rest = sys.argv[2].partition('=')
tune = {}
for item in filter(None, rest.split(',')):
    sw, _, val = item.partition(':')
    tune[sw] = None if val == 'none' else int(val)
f = make()


def engine(t):
    eng = hip.Engine(f, C, k, tune=t)
    eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
    eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
    for _ in range(3):
        lnl = eng.marginal_pass(posterior=False, lh=False)[0]
    return eng, lnl


ballast, _ = engine({})
eng, lnl = engine(tune)
passes, sweeps = [], []
reps = 20 if f.n_nodes > 100000 else 200
for rnd in range(3):
    eng.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        eng.marginal_pass(posterior=False, lh=False)
    eng.sync(); passes.append((time.perf_counter() - t0) / reps * 1e3)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.bottom_up(True)
    eng.sync(); sweeps.append((time.perf_counter() - t0) / reps * 1e3)
import hashlib
print('%-10s %-12s marginal pass %s ms (min %.3f)   bottom-up %s ms (min %.3f)   ln L %s'
      % (sys.argv[1], name, ' '.join('%.3f' % v for v in passes), min(passes), ' '.join('%.3f' % v for v in sweeps), min(sweeps),
         hashlib.sha256(np.ascontiguousarray(lnl).tobytes()).hexdigest()[:8]), flush=True)
eng.close(); ballast.close()
