"""Marginal pass / bottom-up sweep of the ragged and balanced 262 144-tip trees (32 characters): the numbers of VERDICT r03 item 1."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest

which = sys.argv[1:] or ['ragged64', 'ragged4', 'poly64', 'balanced64']
cases = dict(ragged64=('ragged262k k64 C32', lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 64, 32),
             ragged4=('ragged262k k4 C32', lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 4, 32),
             ragged20=('ragged262k k20 C16', lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 20, 16),
             poly64=('polytomies100k k64 C16', lambda: FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 64, 16),
             balanced64=('balanced2^18 k64 C32', lambda: synthetic.balanced_forest(18), 64, 32),
             balanced8=('balanced2^18 k8 C32', lambda: synthetic.balanced_forest(18), 8, 32),
             ragged8=('ragged262k k8 C32', lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 8, 32),
             balanced4=('balanced2^18 k4 C32', lambda: synthetic.balanced_forest(18), 4, 32),
             balanced12=('balanced2^18 k12 C32', lambda: synthetic.balanced_forest(18), 12, 32),
             ragged12=('ragged262k k12 C32', lambda: FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 12, 32))
for key in which:
    name, make, k, C = cases[key]
    f = make()
    with hip.Engine(f, C, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
        eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
        for _ in range(3):
            lnl = eng.marginal_pass(posterior=False, lh=False)[0]
        eng.sync(); t0 = time.perf_counter()
        for _ in range(20):
            eng.marginal_pass(posterior=False, lh=False)
        eng.sync(); ms = (time.perf_counter() - t0) / 20 * 1e3
        tb = time.perf_counter()
        for _ in range(20):
            eng.bottom_up(True)
        eng.sync(); msb = (time.perf_counter() - tb) / 20 * 1e3
        eng.marginal_pass(posterior=False, lh=False)
        post = eng.download_strided(hip.BUF_POSTERIOR, C - 1, 0, 997)
        info = eng.schedule_info()
    import hashlib
    h = hashlib.sha256(np.ascontiguousarray(lnl).tobytes() + np.ascontiguousarray(post).tobytes()).hexdigest()[:12]
    print('%-24s marginal pass %.3f ms, bottom-up %.3f ms  schedule %s  bits %s' % (name, ms, msb, info, h), flush=True)
