#!/usr/bin/env python3
"""
pastml_pipeline end to end at size: a balanced tree of 2^L tips in a newick file, C characters of k states in a table
file; readers, validation, acr() (F81 + MPPA with parameter optimisation) and the result writers, with the time of each
stage (cProfile when PIPELINE_PROFILE is set).  argv: L C k [work_dir]
"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pastml_amd import pipeline  # noqa: E402


def balanced_newick(levels, rng):
    level = ['t%d:%.4f' % (i, rng.uniform(0.01, 0.2)) for i in range(2 ** levels)]
    while len(level) > 1:
        level = ['(%s,%s):%.4f' % (level[i], level[i + 1], rng.uniform(0.01, 0.2)) for i in range(0, len(level), 2)]
    return level[0] + ';'


def main():
    L, C, k = (int(x) for x in sys.argv[1:4])
    work = sys.argv[4] if len(sys.argv) > 4 else tempfile.mkdtemp(prefix='pastml_pipeline_scale_')
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(3)
    nwk, tab = os.path.join(work, 'tree.nwk'), os.path.join(work, 'data.tab')
    with open(nwk, 'w') as f:
        f.write(balanced_newick(L, rng))
    states = np.array(['s%d' % s for s in range(k)])
    pd.DataFrame({'char%d' % c: states[rng.integers(0, k, size=2 ** L)] for c in range(C)},
                 index=['t%d' % i for i in range(2 ** L)]).to_csv(tab, sep='\t', index_label='id')
    pr = cProfile.Profile() if os.environ.get('PIPELINE_PROFILE') else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    res = pipeline.pastml_pipeline(nwk, data=tab, work_dir=os.path.join(work, 'out'))
    if pr:
        pr.disable()
    dt = time.perf_counter() - t0
    sizes = {f: os.path.getsize(os.path.join(work, 'out', f)) for f in sorted(os.listdir(os.path.join(work, 'out')))}
    print('pastml_pipeline: {} tips, {} characters of {} states: {:.2f} s; ln L {}'.format(
        2 ** L, C, k, dt, [round(r['log_likelihood'], 3) for r in res]))
    print('outputs:', {f: '{:.1f} MB'.format(s / 1e6) for f, s in sizes.items()})
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
        print(s.getvalue()[:8000])


if __name__ == '__main__':
    main()
