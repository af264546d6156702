"""
Randomised soak of round 5's schedules and lean units: for seeded random forests / k / columns / masks, the marginal pass of the default
context against contexts with NO_THIN, NO_WIDE_LEAN, the plain level schedule and scaled-down thin ends -- all run the same lane shapes
(those follow k and the forest), so every output must agree bit for bit; and ln L against the oracle.
Test infrastructure (it checks against oracle/), not collected by pytest: python tests/soak_round5.py [n_cases] [first_seed]
"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.tree import FlatForest
from test_gpu_parity import random_masks, random_spec   # noqa: E402
from oracle import pastml_oracle as orc                 # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == 'big'   # (forests of 30 000 - 120 000 tips, the real thin ends; no oracle run)
variants = [('default', {}), ('no_thin', dict(NO_THIN=1)), ('no_lean', dict(NO_WIDE_LEAN=1)),
            ('levels', dict(BLOCK_NODES=0, SMALL_MAX_NODES=0, SMALL_MANY_NODES=0, NO_SUPER=1, NO_THIN=1)),
            ('small_thin', dict(BLOCK_NODES=0, SMALL_MAX_NODES=0, SMALL_MANY_NODES=0, NO_SUPER=1, THIN_UNITS=200, THIN_BLOCK_NODES=16,
                                NARROW_UNITS=8))]
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    tips = int(rng.choice([30000, 60000, 120000])) if big else int(rng.choice([60, 300, 1200, 4000, 9000]))
    arity = int(rng.choice([2, 2, 3, 4, 6]))
    flat = FlatForest.random(tips, seed=seed0 + case, max_arity=arity, n_trees=int(rng.integers(1, 4)),
                             zero_frac=float(rng.choice([0.0, 0.0, 0.05])))
    k = int(rng.choice([2, 3, 4, 5, 8, 12, 16, 17, 20, 32, 33, 48, 64, 65, 100, 130, 256]))
    C = int(rng.integers(4, 17)) if big else int(rng.integers(1, 5))
    if big and k > 64:
        k = int(rng.choice([65, 100]))
    specs = [(random_spec('F81', k, rng), (float(rng.uniform(0.3, 3)), 0.0, 1.0)) for _ in range(C)]
    masks = np.stack([random_masks(flat, k, rng, missing=0.05, multi=0.05, internal=0.02) for _ in range(C)])
    out = {}
    for name, tune in (variants[:3] if big else variants):
        with hip.Engine(flat, C, k, tune=tune, keep_td=bool(case % 2)) as eng:
            eng.set_models(specs)
            eng.set_masks(masks)
            try:
                res = list(eng.marginal_pass())
            except hip.ZeroLikelihoodError as e:
                res = [e.loglik, e.err_parent, e.err_child]
            else:
                res.append(eng.download(hip.BUF_BU, C - 1))
                if case % 2:
                    res.append(eng.download(hip.BUF_TD, 0))
            out[name] = res
    ok = all(len(out[n]) == len(out['default']) and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out['default'], out[n]))
             for n in list(out)[1:])
    lnl = out['default'][0][0]
    try:
        if big:
            raise KeyError
        ref = orc.bottom_up(flat, masks[0].astype(int), specs[0][0], *specs[0][1])
        close = abs(lnl - ref['loglik']) <= 1e-9 * abs(ref['loglik'])
    except KeyError:
        ref, close = dict(loglik=float('nan')), True
    except orc.OracleLikelihoodError:   # (a zero-length branch between conflicting states: the library must have said so too)
        ref = dict(loglik=float('nan'))
        close = len(out['default']) == 3 and out['default'][2][0] >= 0
    if not (ok and close):
        bad += 1
    print('case %3d tips %5d arity %d trees k %3d C %d nodes %6d: variants %s, oracle %s (%.10g / %.10g)'
          % (seed0 + case, tips, arity, k, C, flat.n_nodes, 'same bits' if ok else 'DIFFER', 'ok' if close else 'OFF', lnl, ref['loglik']),
          flush=True)
print('%d cases, %d bad' % (n_cases, bad))
sys.exit(1 if bad else 0)
