#!/usr/bin/env python3
"""Balanced tree, small k, many columns: kernel time of the marginal pass against the schedule's byte model
(argv: levels k C [reps])."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from pastml_amd import hip, synthetic  # noqa: E402

levels, k, C = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
flat = synthetic.balanced_forest(levels)
eng = hip.Engine(flat, C, k)
specs = [dict(kind=0, pi=synthetic.f81_frequencies(k, c)) for c in range(C)]
eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
eng.profile_enable(True)
for i in range(reps + 1):
    if i == 1:
        for w in (0, 1, 2):
            eng.profile_read(w, reset=True)
    eng.set_models([(s, (1.0, 0.0, 1.0)) for s in specs])
    eng.bottom_up(True)
    eng.top_down_marginals(posterior=False, lh=False)
    eng.sync()
sb = bench.schedule_bytes(flat, k, C)
out = {}
for name, w in (('bottom_up', 0), ('top_down', 1), ('prep', 2)):
    ms, n = eng.profile_read(w)
    out[name] = dict(ms=ms / reps, launches=n / reps, model_gb=sb[name] * C / 1e9,
                     tbs=sb[name] * C / (ms / reps * 1e-3) / 1e12 if ms else None)
out['per_unit_bytes'] = sb['per_unit']
out['total_ms'] = sum(out[w]['ms'] for w in ('bottom_up', 'top_down', 'prep'))
print(json.dumps(out))
eng.close()
