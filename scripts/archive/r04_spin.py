"""Host time of a bottom-up sweep on the HIV1C tree with and without the spin on the kernel's completion word
(PASTML_HIP_NO_SPIN_WAIT), alternating; k = 12, 14 columns (the cfg5-shaped gradient of bench.py) and 128 columns."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pastml_amd import hip  # noqa: E402
from pastml_amd.tree import read_tree, get_flat_forest  # noqa: E402

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
flat = get_flat_forest([read_tree(os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])
k = 12
for cols in (14, 128):
    rng = np.random.default_rng(5)
    states = np.tile(rng.integers(0, k, size=flat.n_tips), (cols, 1))
    pis = rng.dirichlet(np.ones(k) * 5, size=cols)
    engines = {name: hip.Engine(flat, cols, k, tune=tune) for name, tune in (('spin', {}), ('sync', dict(NO_SPIN_WAIT=1)))}
    out = {}
    for rep in range(3):
        for name, eng in engines.items():
            eng.set_tip_states(states)
            specs = [(dict(kind=0, pi=pis[c]), (5.5 + 1e-8 * c, 0.0, 1.0)) for c in range(cols)]
            eng.set_models(specs)
            lnl = eng.bottom_up(True)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(300):
                eng.set_models(specs)
                lnl = eng.bottom_up(True)
            eng.sync()
            out.setdefault(name, []).append(((time.perf_counter() - t0) / 300 * 1e3, lnl.copy()))
    assert np.array_equal(out['spin'][0][1], out['sync'][0][1])
    print('cols %3d  ' % cols + '  '.join('%s %s ms' % (n, ' '.join('%.4f' % t for t, _ in v)) for n, v in out.items()), flush=True)
    for eng in engines.values():
        eng.close()
