"""
A/B of schedule switches inside ONE process: engines with different `tune` dictionaries on the same forest, timed alternately
(three rounds of 20 passes each; the boxes and the first launches of a process differ by more than the effects looked for).
usage: tune_ab.py <case> <name=SWITCH:value,SWITCH:value> <name=...> ...     (value 'none' = switch not given)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest

def hiv1c_forest():
    from pastml_amd.tree import read_tree, get_flat_forest
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return get_flat_forest([read_tree(os.path.join(repo, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])


cases = dict(ragged64=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 64, 32),
             ragged12=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 12, 32),
             ragged4=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 4, 32),
             ragged2=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 2, 32),
             poly64=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 64, 16),
             poly4=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 4, 16),
             poly64c32=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 64, 32),
             ragged64c16=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 64, 16),
             poly3_64=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 64, 16),
             poly40=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 40, 16),
             poly3_64c32=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 64, 32),
             poly3_64c8=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 64, 8),
             poly64c8=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 64, 8),
             poly8_64=(lambda: FlatForest.random(100000, seed=9, max_arity=8, n_trees=1), 64, 16),
             bigpoly3_64=(lambda: FlatForest.random(262144, seed=11, max_arity=3, n_trees=1), 64, 32),
             poly20=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 20, 16),
             ragged20=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 20, 32),
             mid64=(lambda: FlatForest.random(40000, seed=7, max_arity=2, n_trees=1), 64, 8),
             balanced64=(lambda: synthetic.balanced_forest(18), 64, 32),
             ragged8=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 8, 32),
             ragged16=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 16, 32),
             ragged32=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 32, 32),
             poly3_20=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 20, 16),
             poly3_32=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 32, 16),
             bin100k_4=(lambda: FlatForest.random(100000, seed=5, max_arity=2, n_trees=2), 4, 16),
             poly3_4=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 4, 16),
             bin100k_12=(lambda: FlatForest.random(100000, seed=5, max_arity=2, n_trees=2), 12, 16),
             poly3_12=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 12, 16),
             bin100k_64=(lambda: FlatForest.random(100000, seed=5, max_arity=2, n_trees=2), 64, 16),
             midpoly3_4=(lambda: FlatForest.random(3600, seed=6, max_arity=3, n_trees=1), 4, 14),
             midpoly5_4=(lambda: FlatForest.random(3600, seed=6, max_arity=5, n_trees=1), 4, 14),
             midpoly3_12=(lambda: FlatForest.random(3600, seed=6, max_arity=3, n_trees=1), 12, 14),
             midpoly5_12=(lambda: FlatForest.random(3600, seed=6, max_arity=5, n_trees=1), 12, 14),
             midpoly5_2=(lambda: FlatForest.random(3600, seed=6, max_arity=5, n_trees=1), 2, 246),
             smallpoly4_5=(lambda: FlatForest.random(300, seed=6, max_arity=4, n_trees=1), 5, 1),
             poly12=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 12, 16),
             balanced4=(lambda: synthetic.balanced_forest(18), 4, 32),
             balanced12=(lambda: synthetic.balanced_forest(18), 12, 32),
             mid4=(lambda: FlatForest.random(40000, seed=7, max_arity=2, n_trees=1), 4, 8),
             hiv12=(lambda: hiv1c_forest(), 12, 14), hiv40=(lambda: hiv1c_forest(), 40, 14), hiv67c204=(lambda: hiv1c_forest(), 67, 204), hiv30c93=(lambda: hiv1c_forest(), 30, 93), hiv64=(lambda: hiv1c_forest(), 64, 64),
             small40=(lambda: FlatForest.random(600, seed=2, max_arity=2, n_trees=1), 40, 8), hiv2=(lambda: hiv1c_forest(), 2, 246), hiv67=(lambda: hiv1c_forest(), 67, 68),
             cfg2=(lambda: synthetic.balanced_forest(16), 4, 1))
make, k, C = cases[sys.argv[1]]
variants = []
for spec in sys.argv[2:]:
    name, _, rest = spec.partition('=')
    tune = {}
    for item in filter(None, rest.split(',')):
        sw, _, val = item.partition(':')
        tune[sw] = None if val == 'none' else int(val)
    variants.append((name, tune))
f = make()
engines = []
# (the first engine of a process lands on memory that runs a few per cent slower: the first variant is built twice, the
# first instance stays as ballast and the second is the one timed)
variants = [('(ballast)', variants[0][1])] + variants + [(variants[0][0] + '-again', variants[0][1])]
for name, tune in variants:
    eng = hip.Engine(f, C, k, tune=tune)
    eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
    eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
    for _ in range(3):
        lnl = eng.marginal_pass(posterior=False, lh=False)[0]
    engines.append((name, eng, lnl, [], []))
for rnd in range(3):
    for name, eng, lnl, passes, sweeps in engines:
        eng.sync(); t0 = time.perf_counter()
        reps = 20 if f.n_nodes > 100000 else 200
        for _ in range(reps):
            eng.marginal_pass(posterior=False, lh=False)
        eng.sync(); passes.append((time.perf_counter() - t0) / reps * 1e3)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.bottom_up(True)
        eng.sync(); sweeps.append((time.perf_counter() - t0) / reps * 1e3)
ref = engines[0][2]
for name, eng, lnl, passes, sweeps in engines:
    if name == '(ballast)':
        eng.close()
        continue
    print('%-10s %-14s marginal pass %s ms (min %.3f)   bottom-up %s ms (min %.3f)   same ln L: %s'
          % (sys.argv[1], name, ' '.join('%.3f' % v for v in passes), min(passes), ' '.join('%.3f' % v for v in sweeps), min(sweeps),
             np.array_equal(lnl, ref)), flush=True)
    eng.close()
