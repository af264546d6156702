"""Signed difference ours - reference of the optimised ln L of every HIV1C column (one batched acr() call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import pandas as pd
from conftest import load_golden, GOLDEN
from pastml_amd.acr import acr
from pastml_amd.tree import read_tree
D = os.path.join(GOLDEN, 'data', 'hiv1c')
z = load_golden('hiv1c_all')
tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
df.index = df.index.map(str)
np.random.seed(239)
t0 = time.perf_counter()
res = acr(tree, df, prediction_method='MPPA', model='F81')
print('acr: %.2f s' % (time.perf_counter() - t0))
rows = []
for ci, (col, r) in enumerate(zip(df.columns, res)):
    if not z['done'][ci]:
        continue
    ref = float(z['c%d_loglik' % ci])
    d = (r['log_likelihood'] - ref) / abs(ref)
    rows.append((d, col, int(z['n_states'][ci]), r['log_likelihood'], ref, r['model'].sf, float(z['c%d_sf' % ci])))
rows.sort()
print('columns with |ours - ref| > 1e-6 |ref| (signed, ours - ref; positive = our optimum is the better one):')
for d, col, k, a, b, sf, rsf in rows:
    if abs(d) > 1e-6:
        print('  %-12s k=%-3d rel %+.3e  ours %.9f ref %.9f  sf %.6f / %.6f' % (col, k, d, a, b, sf, rsf))
print('worst (ours worse): %+.3e   best (ours better): %+.3e   columns within 1e-9: %d of %d'
      % (rows[0][0], rows[-1][0], sum(abs(r[0]) <= 1e-9 for r in rows), len(rows)))
