#!/usr/bin/env python3
"""
Sanity sweep of the schedule heuristics (subtree blocks, single-launch sweeps for many columns): marginal pass and
bottom-up sweep times of the default schedule against the plain level schedule over a grid of tree shapes, state counts
and column counts.  Prints every case and flags those where the default loses more than 10 %.
    python scripts/schedule_sweep.py [TREES]    # the whole grid (spawns one process per case and schedule); TREES e.g. 14,hiv1c
    python scripts/schedule_sweep.py one TREE K C
"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one(tree, k, C):
    import numpy as np
    from pastml_amd import hip, synthetic
    from pastml_amd.tree import read_tree, get_flat_forest
    if tree == 'hiv1c':
        flat = get_flat_forest([read_tree(os.path.join(REPO, 'tests', 'golden', 'data', 'hiv1c', 'pastml_phyml_tree.nwk'))])
    else:
        flat = synthetic.balanced_forest(int(tree))
    rng = np.random.default_rng(1)
    with hip.Engine(flat, C, k) as eng:
        eng.set_tip_states(rng.integers(0, k, size=(C, flat.n_tips)))
        specs = [(dict(kind=0, pi=rng.dirichlet(np.ones(k) * 4)), (1.0, 0.0, 1.0)) for _ in range(C)]
        eng.set_models(specs)

        def timed(fn, reps):
            fn(); eng.sync(); t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            eng.sync()
            return (time.perf_counter() - t0) / reps * 1e3
        reps = 30 if flat.n_nodes * C * k < 5e7 else 5
        print(json.dumps(dict(bu=timed(lambda: eng.bottom_up(True), reps),
                              mp=timed(lambda: eng.marginal_pass(posterior=False, lh=False), reps))))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        return one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    level = dict(os.environ, PASTML_HIP_BLOCK_NODES='0', PASTML_HIP_SMALL_MANY_NODES='0')
    worst = []
    trees = sys.argv[1].split(',') if len(sys.argv) > 1 else ('10', '12', '14', '16', 'hiv1c')
    for tree in trees:
        for k in (2, 4, 12, 20, 64):
            for C in (1, 4, 16, 64, 256):
                n = 7237 if tree == 'hiv1c' else 2 ** (int(tree) + 1)
                if n * C * k > 3e8:
                    continue
                res = []
                for env in (os.environ, level):
                    out = subprocess.run([sys.executable, __file__, 'one', tree, str(k), str(C)], env=env,
                                         capture_output=True, text=True)
                    res.append(json.loads(out.stdout.strip().splitlines()[-1]))
                rb, rm = res[0]['bu'] / res[1]['bu'], res[0]['mp'] / res[1]['mp']
                flag = ' <-- default slower' if max(rb, rm) > 1.1 else ''
                print('tree {:>5} k {:>2} C {:>3}: bottom-up {:.3f} ms (levels {:.3f}, x{:.2f})  marginal pass {:.3f} ms '
                      '(levels {:.3f}, x{:.2f}){}'.format(tree, k, C, res[0]['bu'], res[1]['bu'], rb, res[0]['mp'],
                                                          res[1]['mp'], rm, flag), flush=True)
                if flag:
                    worst.append((tree, k, C, rb, rm))
    print('cases where the default schedule loses more than 10 %:', worst)


if __name__ == '__main__':
    main()
