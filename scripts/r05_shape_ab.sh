#!/bin/bash
: > gpurun_out/r05z_k67.txt
python - <<'PY' >> gpurun_out/r05z_k67.txt 2>&1
import sys, time; sys.path.insert(0, '.')
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import read_tree, get_flat_forest
f = get_flat_forest([read_tree('tests/golden/data/hiv1c/pastml_phyml_tree.nwk')])
for k, C in ((64, 68), (67, 68), (128, 68), (30, 31), (36, 37)):
    with hip.Engine(f, C, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
        eng.set_tip_states(np.stack([synthetic.tip_states(f.n_tips, k, c) for c in range(C)]))
        for _ in range(3): eng.bottom_up(True)
        eng.sync(); t0 = time.perf_counter()
        for _ in range(200): eng.bottom_up(True)
        eng.sync(); ms = (time.perf_counter() - t0) / 200 * 1e3
        print('hiv tree k=%d C=%d bottom-up %.3f ms  schedule %s' % (k, C, ms, eng.sweep_schedule()), flush=True)
PY
cat gpurun_out/r05z_k67.txt
