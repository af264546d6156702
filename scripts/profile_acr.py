#!/usr/bin/env python3
"""cProfile of acr() on the HIV1C fixture (cfg5): where the host time of an optimised run goes."""
import cProfile
import os
import pstats
import sys
import time

import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd.acr import acr  # noqa: E402
from pastml_amd.tree import read_tree  # noqa: E402

HIV = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c')
column = sys.argv[1] if len(sys.argv) > 1 else 'Loc'


def inputs():
    tree = read_tree(os.path.join(HIV, 'pastml_phyml_tree.nwk'))
    df = pd.read_csv(os.path.join(HIV, 'metadata_subset.tab'), sep='\t', index_col=0, header=0)
    df.index = df.index.map(str)
    return tree, df[[column]].copy()


tree, df = inputs()
acr(tree, df, prediction_method='MPPA', model='F81')  # warm-up (library load, context)
tree, df = inputs()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
res = acr(tree, df, prediction_method='MPPA', model='F81')[0]
pr.disable()
print('acr wall', time.perf_counter() - t0, 'lnL', res['log_likelihood'])
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
