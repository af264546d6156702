"""Thin-level workloads for kernel traces: bottom-up sweeps on the HIV1C tree (k = 4, 12; 14 and 128 columns: subtree
blocks + top / the whole sweep in one launch) and the cfg2 marginal pass.  Prints host times; run under rocprofv3 by
scripts/r04_thin_ab.sh for kernel durations (REPS launches per configuration, in this order)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.tree import read_tree, get_flat_forest  # noqa: E402

REPS = 100
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
flat = get_flat_forest([read_tree(os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])
for k in (4, 12):
    for cols in (14, 128):
        rng = np.random.default_rng(5)
        states = rng.integers(0, k, size=(cols, flat.n_tips))
        states[:, rng.random(flat.n_tips) < 0.1] = -1
        pis = rng.dirichlet(np.ones(k) * 5, size=cols)
        with hip.Engine(flat, cols, k) as eng:
            eng.set_tip_states(states)
            specs = [(dict(kind=0, pi=pis[c]), (5.5 + 1e-8 * c, 0.0, 1.0)) for c in range(cols)]
            eng.set_models(specs)
            lnl = eng.bottom_up(True)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(REPS - 1):
                eng.set_models(specs)
                eng.bottom_up(True)
            eng.sync()
            print('hiv1c k %2d cols %3d  %.4f ms per sweep (host)  lnL[0] %.17g' % (k, cols, (time.perf_counter() - t0) / (REPS - 1) * 1e3, lnl[0]), flush=True)
flat = synthetic.balanced_forest(16)
with hip.Engine(flat, 1, 4) as eng:
    eng.set_tip_states(synthetic.tip_states(flat.n_tips, 4, 0))
    spec = [(dict(kind=0, pi=np.ones(4) / 4), (1.0, 0.0, 1.0))]
    eng.set_models(spec)
    lnl = eng.marginal_pass(posterior=False, lh=False)[0]
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(REPS - 1):
        eng.set_models(spec)
        eng.marginal_pass(posterior=False, lh=False)
    eng.sync()
    print('cfg2  %.4f ms per pass (host)  lnL %.17g' % ((time.perf_counter() - t0) / (REPS - 1) * 1e3, lnl[0]), flush=True)
