#!/usr/bin/env python3
"""
acr() end to end at size: a balanced tree of 2^L tips read from its newick string, C characters of k states in a pandas
table, F81 + MPPA with parameter optimisation -- wall time per stage (argv: L C k).
"""
import os
import sys
import time

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd.acr import acr  # noqa: E402
from pastml_amd.tree import read_tree  # noqa: E402
from pastml_amd import batch  # noqa: E402

L, C, k = (int(x) for x in sys.argv[1:4])


def balanced_newick(levels, rng):
    level = ['t%d:%.4f' % (i, rng.uniform(0.01, 0.2)) for i in range(2 ** levels)]
    while len(level) > 1:
        level = ['(%s,%s):%.4f' % (level[i], level[i + 1], rng.uniform(0.01, 0.2)) for i in range(0, len(level), 2)]
    return level[0] + ';'


rng = np.random.default_rng(3)
t0 = time.time()
nwk = balanced_newick(L, rng)
t1 = time.time()
tree = read_tree(nwk)
t2 = time.time()
# characters that evolve along the tree would be nicer; independent uniform tips are the worst case for the optimiser
states = np.array(['s%d' % s for s in range(k)])
df = pd.DataFrame({'char%d' % c: states[rng.integers(0, k, size=2 ** L)] for c in range(C)},
                  index=['t%d' % i for i in range(2 ** L)])
t3 = time.time()
if os.environ.get('ACR_PROFILE'):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
res = acr(tree, df, prediction_method='MPPA', model='F81')
t4 = time.time()
if os.environ.get('ACR_PROFILE'):
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
print('tips %d, %d characters of %d states: newick %.1f s, read_tree %.1f s, table %.1f s, acr() %.1f s' % (
    2 ** L, C, k, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
print('stats', getattr(batch.run_tasks, 'last_stats', None))
print('lnL', [round(r['log_likelihood'], 3) for r in res][:4])
