#!/usr/bin/env python3
"""Two-level units on / off (PASTML_HIP_NO_SUPER) on the cfg4 shard: ms per step, kernel-time slots, agreement.
argv: [levels=20] [k=64] [C=32] [steps=10]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import hip, synthetic  # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
flat = synthetic.balanced_forest(levels)
specs = [(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)]
states = np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)])
res = {}
for mode in ('off', 'on', 'off', 'on'):
    if mode == 'off':
        os.environ['PASTML_HIP_NO_SUPER'] = '1'
    else:
        os.environ.pop('PASTML_HIP_NO_SUPER', None)
    with hip.Engine(flat, C, k) as eng:
        eng.set_models(specs)
        eng.set_tip_states(states)
        for _ in range(3):
            eng.set_models(specs)
            lnl = eng.marginal_pass(posterior=False, lh=False)[0]
        eng.profile_enable(True)
        for w in range(5):
            eng.profile_read(w, reset=True)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.set_models(specs)
            lnl = eng.marginal_pass(posterior=False, lh=False)[0]
        eng.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        slots = [eng.profile_read(w) for w in range(5)]
        eng.profile_enable(False)
        stride = 4099
        post = eng.download_strided(hip.BUF_POSTERIOR, C - 1, 0, stride)
        lhs = eng.download_strided(hip.BUF_LH_SUM, C - 1, 0, stride)
        lsf = eng.download_strided(hip.BUF_LH_SF, C - 1, 0, stride)
    print(mode, 'ms/step %.3f' % dt, ' '.join('%s %.3f/%d' % (n, ms / steps, la // steps) for n, (ms, la) in
                                               zip(('bu', 'td', 'prep', 'td2', 'bu2'), slots)), flush=True)
    res.setdefault(mode, (lnl, post, lhs, lsf))
a, b = res['off'], res['on']
print('lnl max rel diff', np.max(np.abs(a[0] - b[0]) / np.abs(a[0])))
print('posterior max abs diff', np.nanmax(np.abs(a[1] - b[1])), 'rows sum', np.abs(b[1].sum(axis=1) - 1).max())
print('lh_sum rel', np.max(np.abs(a[2] - b[2]) / np.abs(a[2])), 'sf equal', np.array_equal(a[3], b[3]))
