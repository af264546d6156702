"""The optimiser path of HIV1C column 'Year' (k = 30): ours (batch.TRACE) against the reference's (tests/golden/hiv1c_year_trace.npz)."""
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
tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
df.index = df.index.map(str)
polish = int(os.environ.get('PASTML_AMD_POLISH', 0))
batch.TRACE = {}
np.random.seed(239)
res = acr(tree, df[['Year']].copy(), prediction_method='MPPA', model='F81')[0]
runs = batch.TRACE['Year']
print('ours: ln L %.9f  sf %.6f   reference: ln L %.9f  sf %.6f   (ours - ref) / |ref| = %+.3e'
      % (res['log_likelihood'], res['model'].sf, float(z['loglik']), float(z['sf']),
         (res['log_likelihood'] - float(z['loglik'])) / abs(float(z['loglik']))))
for i, r in enumerate(runs):
    print('our run %d: %d parameters, %d iterations, %d evaluations, f = %.9f, success %s' % (i, len(r['x0']), r['nit'], r['nfev'], r['fun'], r['success']))
for i in range(int(z['n_runs'])):
    print('ref run %d: %d parameters, %d iterations, %d evaluations, f = %.9f, %s' % (i, len(z['run%d_x0' % i]), int(z['run%d_nit' % i]), int(z['run%d_nfev' % i]), float(z['run%d_fun' % i]), str(z['run%d_message' % i])))
# stage 2 (all parameters): where do the paths part?
ours = next(r for r in runs if len(r['x0']) == 30)
ref_it = z['run1_iterates']
n = min(len(ours['iterates']), len(ref_it))
rel = np.array([np.max(np.abs(ours['iterates'][i] - ref_it[i]) / np.maximum(np.abs(ref_it[i]), 1e-300)) for i in range(n)])
print('start points equal: %s' % np.array_equal(ours['x0'], z['run1_x0']))
for thr in (1e-12, 1e-9, 1e-6, 1e-3):
    idx = np.flatnonzero(rel > thr)
    print('first iterate whose parameters differ by more than %g relative: %s' % (thr, idx[0] + 1 if len(idx) else 'none'))
print('f along our path   (every 10th iterate):', ' '.join('%.4f' % v for v in ours['values'][::10]))
print('iterates: ours %d, reference %d; our last 5 values: %s' % (len(ours['iterates']), len(ref_it), ' '.join('%.6f' % v for v in ours['values'][-5:])))
