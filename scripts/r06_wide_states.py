"""Marginal pass and joint pass beyond 256 states (64 lanes x 8 states, plain level schedule) next to k = 256 (64 x 4):
balanced 65 536-tip tree x 4 characters.  GB/s: SURVEY 8d's 48 B per (node, state, character) over the pass."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pastml_amd import hip, synthetic

flat = synthetic.balanced_forest(16)
C = 4
for k in (128, 256, 257, 300, 384, 512):
    rng = np.random.default_rng(k)
    specs = [(dict(kind=0, pi=rng.dirichlet(np.ones(k) * 2)), (1.0, 0.0, 1.0)) for _ in range(C)]
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(specs)
        eng.set_tip_states(np.stack([rng.integers(0, k, size=flat.n_tips).astype(np.int32) for _ in range(C)]))
        for what, fn in (('marginal', lambda: eng.marginal_pass(posterior=False, lh=False)), ('joint', lambda: eng.joint_pass(copy_out=False))):
            for _ in range(3):
                fn()
            hip.device_sync()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                fn()
            hip.device_sync()
            ms = (time.perf_counter() - t0) / n * 1e3
            gbs = flat.n_nodes * k * C * 48 / (ms * 1e-3) / 1e9
            print('k = {:3d}  {:8s} {:7.3f} ms per pass'.format(k, what, ms) + ('   {:6.0f} GB/s at 48 B per node, state and character'.format(gbs) if what == 'marginal' else ''), flush=True)
