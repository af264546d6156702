#!/usr/bin/env python3
"""cProfile of one acr() over all 91 HIV1C columns (cfg5 end to end), second call of the process."""
import cProfile, os, pstats, sys, time
import numpy as np
import pandas as pd
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd.acr import acr  # noqa: E402
from pastml_amd.batch import run_tasks  # noqa: E402
from pastml_amd.tree import read_tree  # noqa: E402
D = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c')


def inputs():
    tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
    df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
    df.index = df.index.map(str)
    return tree, df


tree, df = inputs()
np.random.seed(239)
acr(tree, df, prediction_method='MPPA', model='F81')
for rep in range(2):
    tree, df = inputs()
    np.random.seed(239)
    t0 = time.perf_counter()
    acr(tree, df, prediction_method='MPPA', model='F81')
    print('acr wall %.3f s' % (time.perf_counter() - t0), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in run_tasks.last_stats.items()})
tree, df = inputs()
np.random.seed(239)
pr = cProfile.Profile()
pr.enable()
acr(tree, df, prediction_method='MPPA', model='F81')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(40)
