"""
Helper of tests/test_gpu_multi.py: one rank of a 2-rank acr() run.  Every rank has the whole tree and table; acr()
takes the rank's block of the characters; the summed log-likelihood is reduced by pastml_amd.acr.total_log_likelihood.
Usage: python _acr_rank_worker.py OUT_PREFIX
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np  # noqa: E402

from pastml_amd import sharding  # noqa: E402
from pastml_amd.acr import acr, total_log_likelihood  # noqa: E402
from pastml_amd.tree import read_tree  # noqa: E402


def main():
    from test_gpu_api import TREE_NWK, _multi_character_table
    os.environ['PASTML_HIP_DEVICE'] = os.environ.get('PASTML_TEST_DEVICE', '0')
    comm = sharding.init(device=0)
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)
    res = acr(tree, df, prediction_method='MPPA', model='F81')
    total = total_log_likelihood(res)
    with open('{}{}.json'.format(sys.argv[1], comm.rank), 'w') as f:
        json.dump(dict(rank=comm.rank, characters=[r['character'] for r in res],
                       loglik=[r['log_likelihood'] for r in res], total=total), f)
    comm.barrier()
    sharding.shutdown()


if __name__ == '__main__':
    main()
