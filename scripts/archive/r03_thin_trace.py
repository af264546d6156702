#!/usr/bin/env python3
"""cfg2 marginal pass and the cfg5-shaped gradient, a few repetitions each (for a rocprofv3 kernel trace)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.tree import read_tree, get_flat_forest  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'both'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which in ('cfg2', 'both'):
    flat = synthetic.balanced_forest(16)
    with hip.Engine(flat, 1, 4) as eng:
        spec = dict(kind=0, pi=np.ones(4) / 4)
        eng.set_tip_states(synthetic.tip_states(flat.n_tips, 4, 0))
        for _ in range(reps):
            eng.set_models([(spec, (1.0, 0.0, 1.0))])
            eng.marginal_pass(posterior=False, lh=False)
        eng.sync()
if which in ('cfg5', 'both'):
    flat = get_flat_forest([read_tree(os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])
    k, cols = 12, 14
    rng = np.random.default_rng(5)
    with hip.Engine(flat, cols, k) as eng:
        eng.set_tip_states(np.tile(rng.integers(0, k, size=flat.n_tips), (cols, 1)))
        pis = rng.dirichlet(np.ones(k) * 5, size=cols)
        for _ in range(reps):
            eng.set_models([(dict(kind=0, pi=pis[c]), (5.5 + 1e-8 * c, 0.0, 1.0)) for c in range(cols)])
            eng.bottom_up(True)
        eng.sync()
