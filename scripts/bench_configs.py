#!/usr/bin/env python3
"""
Timings of the other BASELINE.json configs (not the driver's headline bench): cfg2 (65 536 tips, JC k=4, marginal),
cfg3 (262 144 tips, JTT k=20, joint), small-tree sweep latency (Albania-sized) and acr() wall time on Albania.
Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from pastml_amd import hip, synthetic  # noqa: E402
from pastml_amd.tree import FlatForest  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    out = {}
    # ---- cfg2
    flat = synthetic.balanced_forest(16)
    eng = hip.Engine(flat, 1, 4)
    spec = dict(kind=0, pi=np.ones(4) / 4)
    eng.set_models([(spec, (1.0, 0.0, 1.0))])
    eng.set_tip_states(synthetic.tip_states(flat.n_tips, 4, 0))

    def cfg2():
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
        eng.sync()
    ms = timed(cfg2, 20)
    out['cfg2_marginal_ms'] = ms
    out['cfg2_units_per_s'] = flat.n_nodes * 4 / (ms * 1e-3)
    out['cfg2_bu_only_ms'] = timed(lambda: (eng.set_models([(spec, (1.0, 0.0, 1.0))]), eng.bottom_up(True)), 20)
    eng.close()

    # ---- cfg3
    from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX
    from pastml_amd.models.generator import get_diagonalisation
    flat = synthetic.balanced_forest(18)
    d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
    spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
    eng = hip.Engine(flat, 1, 20)
    eng.set_models([(spec, (1.0, 0.0, 1.0))])
    eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
    eng.profile_enable(True)

    def cfg3():
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.bottom_up(False)
        eng.joint_backtrace(copy_out=False)
        eng.sync()
    ms = timed(cfg3, 5)
    out['cfg3_joint_ms'] = ms
    out['cfg3_units_per_s'] = flat.n_nodes * 20 / (ms * 1e-3)
    for w, name in ((0, 'bu'), (2, 'pij')):
        t, n = eng.profile_read(w, reset=True)
        out['cfg3_{}_kernel_ms'.format(name)] = t / 6

    def cfg3m():
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.bottom_up(True)
        eng.top_down_marginals(posterior=False, lh=False)
        eng.sync()
    out['cfg3_marginal_ms'] = timed(cfg3m, 5)
    eng.close()

    # ---- small-tree sweep latency (the optimiser's inner loop): Albania-sized random tree, k=5
    flat = FlatForest.random(154, seed=1, max_arity=2)
    rng = np.random.default_rng(0)
    spec = dict(kind=0, pi=rng.dirichlet(np.ones(5)))
    eng = hip.Engine(flat, 1, 5)
    eng.set_tip_states(rng.integers(0, 5, size=flat.n_tips))
    out['small_tree_levels'] = int(flat.n_bu_levels)

    def small():
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.bottom_up(True)
    out['small_tree_bu_sweep_us'] = timed(small, 200) * 1e3
    eng.close()

    # ---- HIV1C-sized random tree, k=12
    flat = FlatForest.random(3619, seed=2, max_arity=2)
    spec = dict(kind=0, pi=rng.dirichlet(np.ones(12)))
    eng = hip.Engine(flat, 1, 12)
    eng.set_tip_states(rng.integers(0, 12, size=flat.n_tips))
    out['hiv_sized_levels'] = int(flat.n_bu_levels)

    def mid():
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.bottom_up(True)
    out['hiv_sized_bu_sweep_us'] = timed(mid, 200) * 1e3
    eng.close()

    # ---- acr() on Albania
    import pandas as pd
    from pastml_amd.acr import acr
    from pastml_amd.tree import read_tree
    data = os.path.join(REPO, 'tests', 'golden', 'data')
    df = pd.read_csv(os.path.join(data, 'data.txt'), index_col=0, header=0)[['Country']]
    for _ in range(2):
        tree = read_tree(os.path.join(data, 'Albanian.tree.152tax.tre'))
        t0 = time.perf_counter()
        res = acr(tree, df, prediction_method='MPPA', model='F81')[0]
        out['albania_acr_mppa_f81_s'] = time.perf_counter() - t0
    out['albania_loglik'] = res['log_likelihood']
    print(json.dumps(out))


if __name__ == '__main__':
    main()
