"""Bits of the sweeps with more than 64 states: hashes of ln L, posteriors, sums, scales, bottom-up and top-down vectors -- run once with
the library in place and once with PASTML_HIP_LIBRARY = a build of the sources before the multi-word lean units (the sequential path)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest, read_tree, get_flat_forest

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = lambda *a: hashlib.sha256(b''.join(np.ascontiguousarray(x).tobytes() for x in a)).hexdigest()[:12]
rng = np.random.default_rng(7)
forests = [('hiv1c', get_flat_forest([read_tree(os.path.join(repo, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])),
           ('random binary 3000', FlatForest.random(3000, seed=3, max_arity=2, n_trees=2)),
           ('polytomies 2500', FlatForest.random(2500, seed=4, max_arity=4, n_trees=1)),
           ('balanced 2^11', synthetic.balanced_forest(11))]
for name, f in forests:
    for k in (65, 67, 100, 128, 130, 200, 256):
        C = 3
        pis = [rng.dirichlet(np.ones(k)) for _ in range(C)]
        masks = np.zeros((C, f.n_nodes, (k + 63) // 64), dtype=np.uint64)
        full = [(1 << min(64, k - 64 * w)) - 1 for w in range((k + 63) // 64)]
        masks[:] = np.array(full, dtype=np.uint64)
        tips = np.flatnonzero(np.asarray(f.n_children) == 0)
        for c in range(C):
            states = rng.integers(0, k, size=tips.size)
            for i, (t, s) in enumerate(zip(tips, states)):
                u = rng.random()
                if u < 0.05:
                    continue                      # unobserved tip: everything allowed
                masks[c, t, :] = 0
                masks[c, t, s >> 6] = np.uint64(1) << np.uint64(s & 63)
                if u < 0.10:                      # ambiguous: a second state
                    s2 = int(rng.integers(0, k))
                    masks[c, t, s2 >> 6] |= np.uint64(1) << np.uint64(s2 & 63)
        with hip.Engine(f, C, k, keep_td=True) as eng:
            eng.set_models([(dict(kind=0, pi=pis[c]), (1.0 + c, 0.0, 1.0)) for c in range(C)])
            eng.set_mask_words(masks)
            lnl, post, lh_sum, lh_sf = eng.marginal_pass()
            bu = eng.download(hip.BUF_BU, 1)
            td = eng.download(hip.BUF_TD, 1)
            eng.sync(); t0 = time.perf_counter()
            for _ in range(50):
                eng.bottom_up(True)
            eng.sync(); ms = (time.perf_counter() - t0) / 50 * 1e3
        print('%-20s k=%3d  %s  bottom-up %.3f ms' % (name, k, h(lnl, post, lh_sum, lh_sf, bu, td), ms), flush=True)
