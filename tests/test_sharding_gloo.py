"""
CPU test of the N > 1 path: characters sharded over 2 ranks, per-rank log-likelihoods reduced through the
communicator interface of pastml_amd.sharding (here its gloo implementation; on GPUs the library's RCCL one,
tests/test_gpu_multi.py).  The per-rank likelihoods come from the oracle (the checker), since there is no GPU here;
the GPU ranks compute the same numbers (tests/test_gpu_parity.py, tests/test_gpu_multi.py).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO
from pastml_amd.sharding import shard_characters


def test_shard_characters_partitions():
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        got = [list(shard_characters(n, r, w)) for r in range(w)]
        assert sum(got, []) == list(range(n))
        sizes = [len(g) for g in got]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_characters(4, 4, 4)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chars, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      PASTML_AMD_COMM='gloo')
    from oracle import pastml_oracle as orc
    from pastml_amd import synthetic, sharding
    comm = sharding.init()
    assert comm.name == 'gloo' and (comm.rank, comm.world) == (rank, world)
    flat = synthetic.balanced_forest(5)
    k = 6
    mine = []
    for c in shard_characters(n_chars, rank, world):
        masks = synthetic.one_hot_masks(flat, k, synthetic.tip_states(flat.n_tips, k, c)).astype(int)
        mine.append(orc.bottom_up(flat, masks, dict(kind=0, pi=synthetic.f81_frequencies(k, c)))['loglik'])
    total = comm.allreduce_loglik(mine)
    everyone = sharding.gather_floats(mine).tolist()
    slowest = float(comm.allreduce([float(rank)], op='max')[0])
    if rank == 0:
        out.put((total, everyone, slowest))
    comm.barrier()
    sharding.shutdown()


@pytest.mark.parametrize('n_chars,world', [(4, 2), (256, 8)])
def test_loglik_allreduce_over_ranks(n_chars, world):
    """world 2: the smallest split; world 8 x 32 characters: BASELINE config 4's partition (256 characters over the 8 GPUs
    of a node) over gloo on the CPU -- every rank its contiguous block, one all-reduce of the summed ln L."""
    from oracle import pastml_oracle as orc
    from pastml_amd import synthetic
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chars, out)) for r in range(world)]
    for p in procs:
        p.start()
    total, everyone, slowest = out.get(timeout=120)
    assert slowest == world - 1
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    flat = synthetic.balanced_forest(5)
    k = 6
    ref = [orc.bottom_up(flat, synthetic.one_hot_masks(flat, k, synthetic.tip_states(flat.n_tips, k, c)).astype(int),
                         dict(kind=0, pi=synthetic.f81_frequencies(k, c)))['loglik'] for c in range(n_chars)]
    np.testing.assert_allclose(everyone, ref, rtol=1e-14)
    np.testing.assert_allclose(total, sum(ref), rtol=1e-14)


def test_local_communicator_and_id_exchange(tmp_path, monkeypatch):
    from pastml_amd import sharding
    c = sharding.LocalCommunicator()
    assert c.allreduce_loglik([1.5, 2.5]) == 4.0
    assert list(sharding.gather_floats([1.0, 2.0], comm=c)) == [1.0, 2.0]
    monkeypatch.setenv('PASTML_AMD_RDZV_DIR', str(tmp_path / 'rdzv'))
    monkeypatch.setattr(sharding, '_RDZV_SEQ', [0])
    data, path = sharding.exchange_unique_id(0, lambda: b'x' * 128)
    monkeypatch.setattr(sharding, '_RDZV_SEQ', [0])
    got, path2 = sharding.exchange_unique_id(1, None, timeout=5)
    assert got == data == b'x' * 128 and path == path2


def test_communicator_setup_is_all_or_none(tmp_path, monkeypatch):
    """
    sharding.agree: before any rank enters the collective initialisation every rank reports "ready" or its failure in
    the rendezvous directory.  One failing rank makes EVERY rank raise (no rank waits inside ncclCommInitRank, no rank
    falls back to another communicator on its own); a rank that never reports makes the others raise after the bound.
    """
    import threading
    from pastml_amd import sharding
    monkeypatch.setenv('PASTML_AMD_RDZV_DIR', str(tmp_path / 'a'))
    outcome = {}

    def rank(r, ok, stage, timeout=20):
        try:
            sharding.agree(r, 3, ok, reason='no librccl here', stage=stage, timeout=timeout)
            outcome[r] = 'went on'
        except RuntimeError as e:
            outcome[r] = str(e)

    def run(verdicts, stage, ranks=(0, 1, 2), timeout=20):
        outcome.clear()
        threads = [threading.Thread(target=rank, args=(r, verdicts[r], stage, timeout)) for r in ranks]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        return dict(outcome)

    assert run({0: True, 1: True, 2: True}, 'ready0') == {0: 'went on', 1: 'went on', 2: 'went on'}
    got = run({0: True, 1: False, 2: True}, 'ready1')
    assert all('rank 1 failed: no librccl here' in got[r] for r in (0, 1, 2))
    got = run({0: True, 1: True, 2: True}, 'ready2', ranks=(0, 2), timeout=0.5)   # rank 1 never shows up
    assert all('ranks [1] did not report' in got[r] for r in (0, 2))
