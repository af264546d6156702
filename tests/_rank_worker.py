"""
Helper of tests/test_gpu_multi.py: one rank of a 2-rank job.  Computes its shard of the characters through the HIP
path on the GPU it is told to use, reduces the log-likelihoods through pastml_amd.sharding and (rank 0) writes the
result as JSON.  Usage: python _rank_worker.py OUT.json N_CHARS LEVELS K
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

from pastml_amd import hip, sharding, synthetic  # noqa: E402


def main():
    out_path, n_chars, levels, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world, local_rank = sharding.rank_world()
    device = int(os.environ.get('PASTML_TEST_DEVICE', local_rank))
    flat = synthetic.balanced_forest(levels)
    chars = list(sharding.shard_characters(n_chars, rank, world))
    with hip.Engine(flat, len(chars), k, device=device) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in chars])
        eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in chars]))
        comm = sharding.init(device=device, engine=eng)
        lnl = eng.bottom_up(True)
        total = comm.allreduce_loglik(lnl)
        everyone = sharding.gather_floats(lnl)
        slowest = comm.allreduce([float(rank)], op='max')
        comm.barrier()
        if rank == 0:
            with open(out_path, 'w') as f:
                json.dump(dict(total=total, per_char=everyone.tolist(), comm=comm.name, max_rank=float(slowest[0])), f)
        sharding.shutdown()


if __name__ == '__main__':
    main()
