"""A few marginal passes of one case of scripts/r04_ragged.py, for rocprofv3 (direct submission: every launch shows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('PASTML_HIP_NO_GRAPH', '1')
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest

key = sys.argv[1] if len(sys.argv) > 1 else 'ragged64'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cases = dict(ragged64=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 64, 32),
             ragged4=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 4, 32),
             balanced64=(lambda: synthetic.balanced_forest(18), 64, 32),
             ragged12=(lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 12, 32),
             poly3_64=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 64, 16),
             poly3_20=(lambda: FlatForest.random(100000, seed=5, max_arity=3, n_trees=2), 20, 16),
             poly4=(lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 4, 16),
             hiv12=(None, 12, 14), cfg2=(lambda: synthetic.balanced_forest(16), 4, 1))
make, k, C = cases[key]
f = make()
with hip.Engine(f, C, k) as eng:
    eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
    eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
    for _ in range(reps):
        eng.marginal_pass(posterior=False, lh=False)
    eng.sync()
print('done', key, reps)
