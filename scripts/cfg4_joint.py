#!/usr/bin/env python3
"""cfg4 shard (1 048 576 tips, k=64, F81, 32 characters): the joint sweep + back-trace and the device state selection,
i.e. what ml_acr adds to the marginal pass for MPPA with force_joint (ml.py:640-750)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
k = 64
flat = synthetic.balanced_forest(levels)
eng = hip.Engine(flat, C, k)
specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
eng.profile_enable(True)
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    lnl = eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    eng.sync()
    t1 = time.perf_counter()
    lnl_j = eng.bottom_up(False)
    eng.sync()
    t2 = time.perf_counter()
    eng.joint_backtrace(copy_out=False)
    eng.sync()
    t3 = time.perf_counter()
    print('marginal pass ms', (t1 - t0) * 1e3, 'joint BU ms', (t2 - t1) * 1e3, 'back-trace ms', (t3 - t2) * 1e3)
eng.close()
