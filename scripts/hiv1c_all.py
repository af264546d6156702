#!/usr/bin/env python3
"""
BASELINE config 5 in full: acr(MPPA, F81, parameter optimisation) on the HIV1C tree (3 619 tips) for every usable
column of the annotation table (91 of them; fixture tests/golden/data/hiv1c/metadata_all.tab.gz) in ONE call -- the
characters are batched as device columns.  Prints wall time and, if the reference goldens are present, the largest
deviation of the optimised log-likelihoods.  Usage: hiv1c_all.py [max_k] [column ...]
"""
import json
import os
import sys
import time

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd.acr import acr  # noqa: E402
from pastml_amd.batch import run_tasks  # noqa: E402
from pastml_amd.tree import read_tree  # noqa: E402

D = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c')
max_k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
only = sys.argv[2:]
tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0)
df.index = df.index.map(str)
ks = {c: len([_ for _ in df[c].unique() if not pd.isna(_) and '' != _]) for c in df.columns}
cols = [c for c in df.columns if ks[c] <= max_k and (not only or c in only)]
np.random.seed(239)
t0 = time.perf_counter()
res = acr(tree, df[cols].copy(), prediction_method='MPPA', model='F81')
dt = time.perf_counter() - t0
out = dict(columns=len(cols), seconds=dt, stats=run_tasks.last_stats,
           loglik={r['character']: r['log_likelihood'] for r in res})
gpath = os.path.join(REPO, 'tests', 'golden', 'hiv1c_all.npz')
if os.path.exists(gpath):
    z = np.load(gpath)
    names = list(z['columns'])
    worst = 0.0
    for r in res:
        ci = names.index(r['character'])
        key = 'c{}_loglik'.format(ci)
        if key in z:
            d = r['log_likelihood'] - float(z[key])
            out.setdefault('delta', {})[r['character']] = d
            worst = max(worst, abs(d))
    out['max_abs_loglik_delta_vs_reference'] = worst
print(json.dumps(out))
