"""HIV1C 'Year' (k = 30) and the other many-state columns under the search's switches (PASTML_AMD_CONTINUE, _POLISH_STEP)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import pandas as pd
from conftest import load_golden, GOLDEN
from pastml_amd import batch
from pastml_amd.acr import acr
from pastml_amd.tree import read_tree
D = os.path.join(GOLDEN, 'data', 'hiv1c')
z = load_golden('hiv1c_all')
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
df.index = df.index.map(str)
names = list(z['columns'])
wide = [c for c in df.columns if int(z['n_states'][names.index(c)]) >= 20]
cols = wide if len(sys.argv) < 2 or sys.argv[1] != 'all' else list(df.columns)
batch.TRACE = {}
tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
np.random.seed(239)
t0 = time.perf_counter()
res = acr(tree, df[cols].copy(), prediction_method='MPPA', model='F81')
dt = time.perf_counter() - t0
print('CONTINUE=%s POLISH_STEP=%s: %d columns in %.2f s' % (os.environ.get('PASTML_AMD_CONTINUE', '1'),
                                                            os.environ.get('PASTML_AMD_POLISH_STEP', '1e-6'), len(cols), dt))
worst = 0.0
for r in res:
    ci = names.index(r['character'])
    if not z['done'][ci]:
        continue
    ref = float(z['c%d_loglik' % ci])
    d = (r['log_likelihood'] - ref) / abs(ref)
    worst = min(worst, d)
    k = int(z['n_states'][ci])
    if k >= 20 or abs(d) > 1e-6:
        runs = batch.TRACE.get(r['character'], [])
        info = ['%d it%s%s' % (q['nit'], '' if not q.get('continued_at') else ' (continued at %d)' % q['continued_at'][0],
                               '' if not q.get('polish') else ' + polish %d it %+.2e' % (q['polish']['nit'], q['fun'] - q['polish']['fun']))
                for q in runs]
        print('  %-12s k=%2d  ln L %.6f  ref %.6f  rel %+.2e   runs: %s' % (r['character'], k, r['log_likelihood'], ref, d, '; '.join(info)))
print('  worst shortfall %.2e' % worst)
