"""HIV1C 'Year': where our search ends when every L-BFGS-B start point is moved by j * 1e-12 relative (j = 0 .. 15)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import pandas as pd
from conftest import load_golden, GOLDEN
from pastml_amd import batch
from pastml_amd.acr import acr
from pastml_amd.tree import read_tree
D = os.path.join(GOLDEN, 'data', 'hiv1c')
z = load_golden('hiv1c_year_trace')
zp = load_golden('hiv1c_year_trace_perturbed')
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
df.index = df.index.map(str)
real = batch.lbfgsb_steps
out = []
for j in range(16):
    eps = j * 1e-12
    batch.lbfgsb_steps = lambda x0, bounds, iterates=None, eps=eps: real(np.asarray(x0, dtype=np.float64) * (1.0 + eps), bounds, iterates)
    batch.TRACE = {}
    tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
    np.random.seed(239)
    res = acr(tree, df[['Year']].copy(), prediction_method='MPPA', model='F81')[0]
    runs = batch.TRACE['Year']
    out.append(res['log_likelihood'])
    print('start points moved by %5.0e: ln L %.6f after %s iterations' % (eps, res['log_likelihood'], [r['nit'] for r in runs]), flush=True)
out = np.array(out)
print('ours: best %.6f, worst %.6f, median %.6f;  the reference: %.6f, and %.6f with its start points moved by 1e-12'
      % (out.max(), out.min(), np.median(out), float(z['loglik']), float(zp['loglik'])))
