"""
Rounding noise of ln L along a line through parameter space, ours (GPU) against the oracle (numpy, the reference's operation
sequence): HIV1C column 'Year' (k = 30) at the reference's optimum, points x + t * 1e-9 (t = 0 .. 40) along the scaling factor
and along one frequency ratio.  In exact arithmetic ln L is a straight line over such a stretch; what is left after a
quadratic fit is the noise the optimiser's forward differences (step 1e-8) divide by 1e-8.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import pandas as pd
from conftest import load_golden, GOLDEN
from oracle import pastml_oracle as orc
from pastml_amd import hip
from pastml_amd.annotation import preannotate_forest
from pastml_amd.batch import annotation_words, masks_from_words
from pastml_amd.tree import read_tree, get_flat_forest
D = os.path.join(GOLDEN, 'data', 'hiv1c')
z = load_golden('hiv1c_year_trace')
tree = read_tree(os.path.join(D, 'pastml_phyml_tree.nwk'))
df = pd.read_csv(os.path.join(D, 'metadata_all.tab.gz'), sep='\t', index_col=0, header=0, dtype=str)
df.index = df.index.map(str)
preannotate_forest([tree], df=df[['Year']])
states = np.array(sorted(s for s in df['Year'].unique() if not pd.isna(s) and s != ''))
k = len(states)
flat = get_flat_forest([tree])
words, _ = annotation_words(flat, 'Year', states)
full = np.zeros_like(words)
masks = masks_from_words(words, k).astype(int)
masks[masks.sum(axis=1) == 0] = 1           # unannotated nodes: every state allowed
sf0, pi0 = float(z['sf']), np.array(z['frequencies'], dtype=np.float64)
T = 41
for label in ('scaling factor', 'frequency ratio 3'):
    pts = []
    for t in range(T):
        if label == 'scaling factor':
            pts.append((pi0, sf0 * (1 + t * 1e-9)))
        else:
            r = pi0 / pi0[-1]
            r = r.copy(); r[3] *= (1 + t * 1e-9)
            pts.append((r / r.sum(), sf0))
    with hip.Engine(flat, T, k) as eng:
        eng.set_models([(dict(kind=0, pi=p), (s, 0.0, 1.0)) for p, s in pts])
        eng.set_masks(np.stack([masks] * T))
        ours = eng.bottom_up(True).copy()
    ref = np.array([orc.bottom_up(flat, masks, dict(kind=0, pi=p), s, 0.0, 1.0, True)['loglik'] for p, s in pts])
    x = np.arange(T, dtype=np.float64)
    for name, y in (('ours (GPU)', ours), ('oracle (numpy)', ref)):
        fit = np.polyval(np.polyfit(x, y - y[0], 2), x)
        res = (y - y[0]) - fit
        print('%-18s along the %-18s: ln L %.9f, slope %.3e per 1e-9, residual rms %.2e, max %.2e' % (name, label, y[0], (y[-1] - y[0]) / (T - 1), res.std(), np.abs(res).max()))
    print('   ours - oracle: mean %.3e, spread %.2e' % ((ours - ref).mean(), (ours - ref).std()))
