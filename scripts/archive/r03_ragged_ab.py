import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest, read_tree
cases = [('ragged262k k64 C32', FlatForest.random(262144, seed=3, max_arity=2, n_trees=1), 64, 32),
         ('ragged262k k4 C32', None, 4, 32), ('ragged262k k20 C16', None, 20, 16),
         ('polytomies100k k64 C16', FlatForest.random(100000, seed=5, max_arity=5, n_trees=2), 64, 16),
         ('balanced2^18 k64 C32', synthetic.balanced_forest(18), 64, 32)]
res = {}
for mode in ('sorted', 'id-order', 'sorted', 'id-order'):
    if mode == 'id-order': os.environ['PASTML_HIP_NO_SHAPE_SORT'] = '1'
    else: os.environ.pop('PASTML_HIP_NO_SHAPE_SORT', None)
    last = None
    for name, f, k, C in cases:
        f = f if f is not None else last
        last = f
        with hip.Engine(f, C, k) as eng:
            eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
            eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
            for _ in range(3):
                lnl = eng.marginal_pass(posterior=False, lh=False)[0]
            eng.sync(); t0 = time.perf_counter()
            for _ in range(10):
                eng.marginal_pass(posterior=False, lh=False)
            eng.sync(); ms = (time.perf_counter() - t0) / 10 * 1e3
            tb = time.perf_counter()
            for _ in range(10):
                eng.bottom_up(True)
            eng.sync(); msb = (time.perf_counter() - tb) / 10 * 1e3
            eng.marginal_pass(posterior=False, lh=False)
            post = eng.download_strided(hip.BUF_POSTERIOR, C - 1, 0, 997)
        print(mode, name, 'marginal pass %.3f ms, bottom-up %.3f ms' % (ms, msb), flush=True)
        key = name
        if key in res:
            assert np.array_equal(res[key][0], lnl) and np.array_equal(res[key][1], post), 'bits differ: ' + name
        else:
            res[key] = (lnl, post)
print('same bits in both orders')
