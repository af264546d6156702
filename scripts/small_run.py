#!/usr/bin/env python3
"""Latency of one bottom-up sweep on an Albania-sized tree (the optimiser's inner loop), split by stage."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip  # noqa: E402
from pastml_amd.tree import FlatForest  # noqa: E402

n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 154
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
flat = FlatForest.random(n_tips, seed=1, max_arity=2)
rng = np.random.default_rng(0)
spec = dict(kind=0, pi=rng.dirichlet(np.ones(k)))
eng = hip.Engine(flat, C, k)
eng.set_tip_states(rng.integers(0, k, size=(C, flat.n_tips)))
models = [(spec, (1.0, 0.0, 1.0))] * C
for _ in range(20):
    eng.set_models(models)
    eng.bottom_up(True)
n = 500
t_set = t_bu = 0.0
for _ in range(n):
    t0 = time.perf_counter()
    eng.set_models(models)
    t1 = time.perf_counter()
    eng.bottom_up(True)
    t2 = time.perf_counter()
    t_set += t1 - t0
    t_bu += t2 - t1
print('levels', flat.n_bu_levels, 'nodes', flat.n_nodes, 'set_models us', t_set / n * 1e6, 'bottom_up us', t_bu / n * 1e6)
eng.close()
