import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from pastml_amd import hip
from pastml_amd.tree import FlatForest
from test_gpu_parity import random_spec, random_masks
k=4
rng = np.random.default_rng(k)
flat = FlatForest.random(300, seed=k + 1, max_arity=3, zero_frac=0.0, n_trees=2)
specs = [random_spec('F81', k, rng) for _ in range(2)]
rates = [(1.3, 0.0, 1.0), (0.8, 0.02, 0.95)]
masks = np.stack([random_masks(flat, k, rng) for _ in range(2)])
out=[]
for fusion in (True, False):
    with hip.Engine(flat, 2, k, cherry_fusion=fusion) as eng:
        eng.set_models(list(zip(specs, rates))); eng.set_masks(masks)
        lnl = eng.bottom_up(True)
        post, lh_sum, lh_sf = eng.top_down_marginals()
        bu = eng.download(hip.BUF_BU, 1); bu_sf = eng.download(hip.BUF_BU_SF, 1)
        out.append((lnl, post, lh_sum, lh_sf, bu, bu_sf))
names=['lnl','post','lh_sum','lh_sf','bu','bu_sf']
for nm,a,b in zip(names,out[0],out[1]):
    d = np.nan_to_num(a)-np.nan_to_num(b)
    print(nm, np.array_equal(a,b,equal_nan=True), np.abs(d).max())
    if not np.array_equal(a,b,equal_nan=True):
        idx=np.argwhere(np.nan_to_num(a)!=np.nan_to_num(b))[:5]; print(idx, [ (a[tuple(i)], b[tuple(i)]) for i in idx])
        if nm in('post','lh_sum','lh_sf'):
            nodes=np.unique(idx[:,1]); print('nodes', nodes[:10], 'is_tip', flat.is_tip[nodes[:10]], 'parent kind: nchildren of parent', flat.n_children[flat.parent[nodes[:10]]])
