#!/usr/bin/env python3
"""Sweeps of the eigen models (JTT, one character) on three trees -- cfg3's balanced tree, HIV1C's tree, a random 40 000-tip
tree: marginal bottom-up sweep, marginal pass, joint sweep, joint pass.  Environment switches select the schedule
(PASTML_HIP_NO_EIGJ_TIERS, PASTML_HIP_NO_EIGG_TIERS, PASTML_HIP_NO_BT_TIERS, PASTML_HIP_EIGJ_TIER_DEPTH / _THIN); TAG labels
the line.  profiles/r03l_eigen_tiers_three_trees.txt was made with this."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.tree import read_tree, FlatForest  # noqa: E402
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX  # noqa: E402
from pastml_amd.models.generator import get_diagonalisation  # noqa: E402

d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)


def timed(eng, fn, reps=50):
    fn()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e3


hiv = os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk')
for name, flat in (('cfg3', synthetic.balanced_forest(18)), ('hiv1c-tree', FlatForest.from_trees([read_tree(hiv)])),
                   ('ragged40k', FlatForest.random(40000, seed=4, max_arity=3, n_trees=1))):
    with hip.Engine(flat, 1, 20) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
        print(os.environ.get('TAG', ''), name,
              'marginal BU %.4f ms, marginal pass %.4f ms, joint sweep %.4f ms, joint pass %.4f ms, lnL %.10f' % (
                  timed(eng, lambda: eng.bottom_up(True)), timed(eng, lambda: eng.marginal_pass(posterior=False, lh=False)),
                  timed(eng, lambda: eng.bottom_up(False)), timed(eng, lambda: eng.joint_pass(copy_out=False)),
                  eng.bottom_up(True)[0]), flush=True)
