"""
GPU tests of the multi-process path: one process per rank, characters sharded, every rank computing its
shard THROUGH THE HIP PATH, the log-likelihoods reduced by pastml_amd.sharding.

The GPU box has one GPU and RCCL refuses two ranks on one device, so the 2-rank test puts both ranks on GPU 0 and
reduces over gloo (PASTML_AMD_COMM=gloo); the library's RCCL communicator (dlopen of librccl, ncclCommInitRank,
ncclAllReduce on the ctx's stream, staging buffers) is exercised for real with a world of one rank
(PASTML_HIP_COMM_FORCE_RCCL=1).  bench.py --gpus 2 is run the same way.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from pastml_amd import hip, sharding, synthetic

pytestmark = pytest.mark.gpu

WORKER = os.path.join(REPO, 'tests', '_rank_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process_logliks(n_chars, levels, k):
    flat = synthetic.balanced_forest(levels)
    with hip.Engine(flat, n_chars, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(n_chars)])
        eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(n_chars)]))
        return eng.bottom_up(True)


def _assert_complete_line(line):
    """An N > 1 line carries what the N = 1 line does (north_star: the CPU path timed in the same run, the roofline)."""
    cpu = line['cpu_baseline']
    assert cpu['value'] > 0 and cpu['cores'] >= 1 and cpu['kind'] == 'port' and cpu['unit'] == line['unit']
    assert line['speedup_vs_cpu_baseline'] == pytest.approx(line['value'] / cpu['value'])
    roof = line['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert roof['achieved'] > 0 and roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert len(line['library']['build_digest']) == 16 and line['library']['build_digest'] == line['library']['source_digest']


def test_two_ranks_on_one_gpu_through_the_hip_path(tmp_path):
    n_chars, levels, k = 6, 10, 64
    port = _free_port()
    out = str(tmp_path / 'rank0.json')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), PASTML_AMD_COMM='gloo', PASTML_TEST_DEVICE='0',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, WORKER, out, str(n_chars), str(levels), str(k)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = json.load(open(out))
    ref = _single_process_logliks(n_chars, levels, k)
    # columns are computed independently and deterministically: the sharded run gives the same bits
    assert got['per_char'] == ref.tolist()
    # the reduced total: each rank's local sum in column order, then rank 0 + rank 1
    local = [0.0, 0.0]
    for r in range(2):
        for c in sharding.shard_characters(n_chars, r, 2):
            local[r] += float(ref[c])
    assert got['total'] == local[0] + local[1]
    assert got['max_rank'] == 1.0 and got['comm'] == 'gloo'


def test_library_rccl_communicator_world_of_one(monkeypatch):
    """dlopen(librccl) + ncclGetUniqueId + ncclCommInitRank + ncclAllReduce on the engine's stream, one rank."""
    monkeypatch.setenv('PASTML_HIP_COMM_FORCE_RCCL', '1')
    flat = synthetic.balanced_forest(6)
    k = 5
    with hip.Engine(flat, 3, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(3)])
        eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(3)]))
        uid = hip.comm_unique_id()
        assert len(uid) == hip.COMM_ID_BYTES and any(uid)
        eng.comm_init(0, 1, uid)
        assert eng.comm_info() == dict(rank=0, world=1, backend='rccl', rccl_ranks=1)   # (ncclCommCount of the communicator)
        lnl = eng.bottom_up(True)
        total = eng.allreduce_loglik(lnl)
        assert total == float(lnl[0]) + float(lnl[1]) + float(lnl[2])
        v = np.array([1.5, -2.0, 1e300, 0.0] * 40)   # grows the staging buffer past its first size
        assert np.array_equal(eng.allreduce(v), v)
        assert np.array_equal(eng.allreduce(v, hip.COMM_MAX), v)
        with pytest.raises(hip.HipError):
            eng.comm_init(0, 1, uid)   # already attached
        eng.comm_destroy()
        with pytest.raises(hip.HipError):
            eng.allreduce([1.0])


def test_communicator_survives_tree_upload_and_plain_world_of_one():
    flat = synthetic.balanced_forest(4)
    with hip.BareContext() as ctx:
        ctx.comm_init(0, 1)
        assert ctx.comm_info() == dict(rank=0, world=1, backend='local', rccl_ranks=0)
        assert len(hip.device_uuid(0)) == 32
        assert ctx.allreduce_loglik([1.0, 2.0, 3.5]) == 6.5
    with pytest.raises(hip.HipError):
        with hip.BareContext() as ctx:
            ctx.comm_init(2, 2)


def test_bench_two_ranks_started_directly(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent starts the ranks itself and relays rank 0's line."""
    env = dict(os.environ, BENCH_ALL_RANKS_ON_GPU0='1', PASTML_AMD_COMM='gloo')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--workload', 'cfg4_small', '--chars-per-gpu', '2', '--cpu-baseline-levels', '11',
                        '--cpu-baseline-cores', '2'], env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['config']['chars_total'] == 4 and line['scaling'] == 'weak'
    _assert_complete_line(line)
    assert line['validation']['columns'] == 2
    ref = _single_process_logliks(4, 14, 64)
    np.testing.assert_allclose(line['loglik_sum'], ref.sum(), rtol=1e-13)


def test_bench_four_ranks_of_32_characters(tmp_path):
    """
    The shard of BASELINE config 4 -- 32 characters per rank -- end to end through `bench.py --gpus N` on a reduced tree
    (16 384 tips): N = 4 ranks on GPU 0 over gloo, 128 characters, every rank validating its 32 columns, the reduced total
    against one process that computes all 128.  (N = 8 x 32 = 256 is what the driver launches on a node; a GPU box admits
    six processes on its card, so the 8-way partition itself is covered on the CPU, tests/test_sharding_gloo.py.)
    """
    env = dict(os.environ, BENCH_ALL_RANKS_ON_GPU0='1', PASTML_AMD_COMM='gloo')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PASTML_AMD_RDZV_DIR'):
        env.pop(key, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4', '--steps', '2', '--warmup', '1',
                        '--workload', 'cfg4_small', '--chars-per-gpu', '32', '--no-secondary', '--cpu-baseline-levels', '11',
                        '--cpu-baseline-cores', '2'], env=env,
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line['n_gpus'] == 4 and line['config']['chars_total'] == 128 and line['scaling'] == 'weak'
    assert line['validation']['columns'] == 32
    _assert_complete_line(line)
    ref = _single_process_logliks(128, 14, 64)
    np.testing.assert_allclose(line['loglik_sum'], ref.sum(), rtol=1e-13)


def test_acr_shards_characters_over_ranks(tmp_path):
    """acr() under a 2-process launch: each rank reconstructs its block of the characters (through the HIP path), the
    total log-likelihood is all-reduced; together the ranks reproduce the single-process run bit for bit."""
    import pandas as pd  # noqa: F401
    from test_gpu_api import TREE_NWK, _multi_character_table
    from pastml_amd.acr import acr
    from pastml_amd.tree import read_tree
    port = _free_port()
    prefix = str(tmp_path / 'rank')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), PASTML_AMD_COMM='gloo', PASTML_TEST_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_acr_rank_worker.py'), prefix], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = [json.load(open('{}{}.json'.format(prefix, r))) for r in range(2)]
    tree = read_tree(TREE_NWK)
    df = _multi_character_table(tree)
    ref = acr(tree, df, prediction_method='MPPA', model='F81')
    assert got[0]['characters'] + got[1]['characters'] == [r['character'] for r in ref]
    assert got[0]['loglik'] + got[1]['loglik'] == [r['log_likelihood'] for r in ref]
    assert got[0]['total'] == got[1]['total']
    np.testing.assert_allclose(got[0]['total'], sum(r['log_likelihood'] for r in ref), rtol=1e-14)


def test_bench_two_ranks_under_torchrun(tmp_path):
    """The driver's way of starting N > 1: torch.distributed.run provides RANK / LOCAL_RANK / WORLD_SIZE; the ranks find
    each other's RCCL id (here: gloo, both ranks on GPU 0) through the launcher-scoped rendezvous directory."""
    env = dict(os.environ, BENCH_ALL_RANKS_ON_GPU0='1', PASTML_AMD_COMM='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PASTML_AMD_RDZV_DIR'):
        env.pop(key, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--workload', 'cfg4_small', '--chars-per-gpu', '2', '--cpu-baseline-levels', '11',
                        '--cpu-baseline-cores', '2'], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1                      # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['chars_total'] == 4 and line['config']['collective'] == 'gloo'
    # who ran where: both ranks report, with the UUID of the one GPU they share
    assert [p['rank'] for p in line['per_rank']] == [0, 1] and all(p['ms_per_step'] > 0 for p in line['per_rank'])
    assert line['per_rank'][0]['device_uuid'] == line['per_rank'][1]['device_uuid'] == hip.device_uuid(0)
    assert line['config']['distinct_devices'] == 1
    _assert_complete_line(line)   # (no parent of ours under a launcher: rank 0 timed the CPU path before touching the GPU)
    np.testing.assert_allclose(line['loglik_sum'], _single_process_logliks(4, 14, 64).sum(), rtol=1e-13)


def test_bench_ends_when_a_rank_dies_before_joining(tmp_path):
    """
    A rank that dies mid-initialisation must not leave the job hanging: `bench.py --gpus 2` watches all its ranks, stops
    the survivor (which waits for the dead rank in the communicator's rendezvous) and exits non-zero -- within seconds,
    not at a collective's timeout.
    """
    import time
    env = dict(os.environ, BENCH_ALL_RANKS_ON_GPU0='1', PASTML_AMD_COMM='gloo', BENCH_TEST_DIE_RANK='1')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PASTML_AMD_RDZV_DIR'):
        env.pop(key, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--workload', 'cfg4_small', '--chars-per-gpu', '1', '--no-cpu-baseline'], env=env,
                       capture_output=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 30
    assert b'rank 1 exited with code 7' in r.stderr


def test_two_ranks_on_one_gpu_over_rccl_fail_loudly(tmp_path):
    """
    The RCCL route with more than one rank cannot succeed on a one-GPU box (RCCL refuses two ranks per device) -- but it
    must FAIL LIKE A JOB SHOULD: ncclCommInitRank really runs with a world of two, its error comes back through
    pml_comm_init with RCCL's own text, every rank exits non-zero and `bench.py --gpus 2` ends within seconds instead of
    hanging in a collective.  (What a wrongly pinned rank on a real 8-GPU node would look like.)
    """
    import time
    env = dict(os.environ, BENCH_ALL_RANKS_ON_GPU0='1', PASTML_AMD_COMM='rccl', PASTML_AMD_RDZV_TIMEOUT='30',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PASTML_AMD_RDZV_DIR'):
        env.pop(key, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--workload', 'cfg4_small', '--chars-per-gpu', '1', '--no-cpu-baseline'], env=env,
                       capture_output=True, timeout=180)
    err = r.stderr.decode()
    assert r.returncode != 0, r.stdout.decode()[-2000:]
    assert time.time() - t0 < 60, err[-3000:]
    assert 'ncclCommInitRank failed' in err, err[-3000:]
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith('{')]   # no result line from a failed job


def test_marginal_pass_reduces_the_total_on_the_device(monkeypatch):
    """
    With a communicator attached pml_marginal_pass puts the one collective of the path on the sweep's stream itself: the
    rank's sum formed on the device (column order), ncclAllReduce, the copy back -- all behind the sweeps and before the
    call's single wait; pml_loglik_total hands the value out.  One rank through librccl (PASTML_HIP_COMM_FORCE_RCCL).
    """
    monkeypatch.setenv('PASTML_HIP_COMM_FORCE_RCCL', '1')
    flat = synthetic.balanced_forest(9)
    k, C = 12, 5
    with hip.Engine(flat, C, k) as eng:
        eng.set_models([(dict(kind=0, pi=synthetic.f81_frequencies(k, c)), (1.0, 0.0, 1.0)) for c in range(C)])
        eng.set_tip_states(np.stack([synthetic.tip_states(flat.n_tips, k, c) for c in range(C)]))
        assert eng.loglik_total() is None                       # no communicator yet
        eng.comm_init(0, 1, hip.comm_unique_id())
        for _ in range(3):                                      # (the third pass replays the captured graph)
            lnl = eng.marginal_pass(posterior=False, lh=False)[0]
            want = 0.0
            for v in lnl:
                want += float(v)
            assert eng.loglik_total() == want
            assert eng.loglik_total() is None                   # handed out once per pass
        comm = sharding.RcclCommunicator.__new__(sharding.RcclCommunicator)
        comm._eng, comm._own, comm.rank, comm.world, comm._rdzv = eng, False, 0, 1, None
        lnl = eng.marginal_pass(posterior=False, lh=False)[0]
        assert comm.allreduce_loglik(lnl) == want               # the communicator takes the device's total
        assert comm.allreduce_loglik(lnl) == want               # ... or reduces the values it is given
        eng.comm_destroy()
