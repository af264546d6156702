"""
Multi-GPU layout of the path: characters (columns) are independent (pastml/acr.py:213-231 runs one ml_acr per
character), so they are sharded over the ranks of one node -- one process per GPU -- with the tree replicated.
No collective touches the data path; the only exchange is the (all-)reduce of the summed log-likelihood, an 8-byte
message over RCCL/xGMI issued by the library itself (``pml_allreduce_loglik``, include/pastml_hip.h) on the stream of
the rank's device context.  The 128-byte RCCL id travels from rank 0 to the others through a file (one node).

Launch: one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK in the environment (``torch.distributed.run`` sets
them, so does ``bench.py --gpus N`` for the processes it starts).  ``init()`` picks the communicator:

* world == 1                       -> :class:`LocalCommunicator` (no library, no GPU needed)
* PASTML_AMD_COMM=gloo             -> :class:`TorchCommunicator` on gloo (CPU: tests and dry runs of the N > 1 path on
                                      boxes with fewer GPUs than ranks -- RCCL refuses two ranks per device)
* otherwise                        -> :class:`RcclCommunicator`.  Whether it can be set up is decided by ALL ranks together
                                      (``agree``): every rank reports "ready" or its failure in the job's rendezvous
                                      directory before anyone enters the collective initialisation; if one rank failed, or
                                      did not report in time, every rank raises -- no rank is left waiting inside
                                      ncclCommInitRank, and there is no per-rank fallback that would split the job over two
                                      communicators.
"""
import os
import time

import numpy as np


def rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def shard_characters(n_chars, rank, world):
    """
    Contiguous block of character indices of a rank: the first n_chars % world ranks get one
    more.  Returns range(begin, end).
    """
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad rank/world {}/{}'.format(rank, world))
    base, extra = divmod(n_chars, world)
    begin = rank * base + min(rank, extra)
    return range(begin, begin + base + (1 if rank < extra else 0))


# ---------------------------------------------------------------------------------------------------------------------
class LocalCommunicator(object):
    """world == 1."""
    name = 'local'
    rank, world = 0, 1

    def allreduce(self, values, op='sum'):
        return np.array(np.atleast_1d(values), dtype=np.float64)

    def allreduce_loglik(self, loglik):
        total = 0.0
        for v in np.atleast_1d(loglik):
            total += float(v)
        return total

    def barrier(self):
        pass

    def close(self):
        pass


_RDZV_SEQ = [0]


def _rendezvous_dir():
    d = os.environ.get('PASTML_AMD_RDZV_DIR')
    if d:
        return d
    # one launcher (torchrun agent, bench.py parent) = one parent process id for all ranks of a job on this node
    return os.path.join(os.environ.get('TMPDIR', '/tmp'),
                        'pastml_amd_rdzv_{}_{}'.format(os.environ.get('MASTER_PORT', '0'), os.getppid()))


def exchange_unique_id(rank, make_id, timeout=300.0):
    """
    Rank 0 calls make_id() and publishes the bytes in the job's rendezvous directory (written under a temporary name,
    then renamed: readers never see a partial file); the other ranks wait for the file.  Successive calls of one job use
    successive file names (every rank calls in the same order).
    """
    d = _rendezvous_dir()
    seq = _RDZV_SEQ[0]
    _RDZV_SEQ[0] += 1
    path = os.path.join(d, 'rccl_id_{}.bin'.format(seq))
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        try:
            data = make_id()
        except Exception:
            with open(path + '.failed', 'w') as f:   # the other ranks must not wait for an id that will not come
                f.write('rank 0 could not create the RCCL id')
            raise
        tmp = path + '.tmp{}'.format(os.getpid())
        with open(tmp, 'wb') as f:
            f.write(data)
        os.replace(tmp, path)
        return data, path
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                data = f.read()
            if data:
                return data, path
        except OSError:
            pass
        if os.path.exists(path + '.failed'):
            raise RuntimeError('rank 0 could not create the RCCL id')
        if time.time() - t0 > timeout:
            raise TimeoutError('rank {}: no RCCL id from rank 0 at {} after {:.0f} s'.format(rank, path, timeout))
        time.sleep(0.01)


def agree(rank, world, ok, reason='', stage='ready', timeout=None):
    """
    All-or-none decision through the rendezvous directory (no collective is available yet): every rank writes its
    verdict, then waits -- for a bounded time, rank 0 included -- until all verdicts are there.  Raises RuntimeError on
    every rank if any rank reported a failure or is missing after ``timeout`` seconds (PASTML_AMD_RDZV_TIMEOUT, 120).
    """
    if timeout is None:
        timeout = float(os.environ.get('PASTML_AMD_RDZV_TIMEOUT', '120'))
    d = _rendezvous_dir()
    os.makedirs(d, exist_ok=True)
    mine = os.path.join(d, '{}_{}'.format(stage, rank))
    tmp = mine + '.tmp{}'.format(os.getpid())
    with open(tmp, 'w') as f:
        f.write('ok' if ok else 'failed: {}'.format(reason))
    os.replace(tmp, mine)
    t0 = time.time()
    verdicts = {}
    while len(verdicts) < world:
        for r in range(world):
            if r not in verdicts:
                try:
                    with open(os.path.join(d, '{}_{}'.format(stage, r))) as f:
                        text = f.read()
                    if text:
                        verdicts[r] = text
                except OSError:
                    pass
        if any(v != 'ok' for v in verdicts.values()):
            break   # no need to wait for the rest
        if len(verdicts) < world:
            if time.time() - t0 > timeout:
                missing = sorted(set(range(world)) - set(verdicts))
                raise RuntimeError('rank {}: ranks {} did not report "{}" within {:.0f} s'.format(rank, missing, stage,
                                                                                                   timeout))
            time.sleep(0.01)
    bad = {r: v for r, v in verdicts.items() if v != 'ok'}
    if bad:
        raise RuntimeError('rank {}: the communicator cannot be set up on every rank: {}'.format(
            rank, '; '.join('rank {} {}'.format(r, v) for r, v in sorted(bad.items()))))


class RcclCommunicator(object):
    """
    The library's communicator on a bare device context of this rank's GPU (``pml_comm_*``, RCCL over xGMI).
    ``engine``: attach to an existing Engine instead (the all-reduce is then ordered on that engine's stream, right
    behind the sweep that produced the values).
    """
    name = 'rccl'

    def __init__(self, rank, world, device=None, engine=None):
        from pastml_amd import hip
        self.rank, self.world = rank, world
        self._own = engine is None
        forced = bool(os.environ.get('PASTML_HIP_COMM_FORCE_RCCL'))
        self._eng = None
        uid = path = None
        problem = None
        seq = _RDZV_SEQ[0]   # successive communicators of one job use successive file names
        self._rdzv = _rendezvous_dir() if world > 1 else None
        try:
            self._eng = engine if engine is not None else hip.BareContext(device)
            if world > 1 or forced:
                uid, path = exchange_unique_id(rank, hip.comm_unique_id,
                                               timeout=float(os.environ.get('PASTML_AMD_RDZV_TIMEOUT', '120')))
        except Exception as e:   # no device, no librccl, no id from rank 0, ...
            problem = e
        if world > 1:
            # everybody, or nobody, enters the collective initialisation
            agree(rank, world, problem is None, reason=repr(problem), stage='ready{}'.format(seq))
        elif problem is not None:
            raise problem
        self._eng.comm_init(rank, world, uid)   # collective: returns once every rank has joined

    def allreduce(self, values, op='sum'):
        from pastml_amd import hip
        return self._eng.allreduce(values, hip.COMM_SUM if op == 'sum' else hip.COMM_MAX)

    def allreduce_loglik(self, loglik):
        # after a marginal pass on the engine this communicator is attached to, the total is already there: the library
        # reduced it on the device, on the sweep's stream (no host round trip of its own)
        total = self._eng.loglik_total() if not self._own else None
        return total if total is not None else self._eng.allreduce_loglik(loglik)

    def barrier(self):
        self._eng.allreduce([0.0])

    def close(self):
        if self._eng is not None:
            self._eng.comm_destroy()
            if self._own:
                self._eng.close()
            self._eng = None
        if self.rank == 0 and self._rdzv:
            # (callers close after a barrier: every rank is past the rendezvous) the job's files can go
            try:
                for name in os.listdir(self._rdzv):
                    os.remove(os.path.join(self._rdzv, name))
                os.rmdir(self._rdzv)
            except OSError:
                pass
            self._rdzv = None


class TorchCommunicator(object):
    """
    torch.distributed's gloo backend (CPU) behind the same interface: tests and dry runs of the N > 1 path on boxes with
    fewer GPUs than ranks.  GPUs talk through the library's own RCCL communicator (:class:`RcclCommunicator`) -- there is
    no second GPU backend.
    """
    name = 'gloo'

    def __init__(self, rank, world):
        import torch.distributed as dist
        self._dist = dist
        self._own = not dist.is_initialized()
        if self._own:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def allreduce(self, values, op='sum'):
        import torch
        t = torch.tensor(np.atleast_1d(values), dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == 'sum' else self._dist.ReduceOp.MAX)
        return t.numpy().copy()

    def allreduce_loglik(self, loglik):
        return float(self.allreduce([LocalCommunicator().allreduce_loglik(loglik)])[0])

    def barrier(self):
        self.allreduce([0.0])

    def close(self):
        if self._own and self._dist.is_initialized():
            self._dist.destroy_process_group()


GlooCommunicator = TorchCommunicator


_COMM = None


def init(device=None, engine=None, kind=None):
    """The process-wide communicator for RANK / WORLD_SIZE / LOCAL_RANK of the environment (idempotent)."""
    global _COMM
    if _COMM is not None:
        return _COMM
    rank, world, local_rank = rank_world()
    kind = kind or os.environ.get('PASTML_AMD_COMM', 'rccl')
    if world == 1 and not os.environ.get('PASTML_HIP_COMM_FORCE_RCCL'):
        _COMM = LocalCommunicator()
    elif kind == 'gloo':
        _COMM = GlooCommunicator(rank, world)
    elif kind != 'rccl':
        raise ValueError('PASTML_AMD_COMM={!r}: the communicators are "rccl" (GPUs) and "gloo" (CPU dry runs)'.format(kind))
    else:
        # no automatic fallback: a failure is a failure of the job, on every rank (RcclCommunicator / agree)
        _COMM = RcclCommunicator(rank, world, device=local_rank if device is None else device, engine=engine)
    return _COMM


def communicator():
    return _COMM


def shutdown():
    global _COMM
    if _COMM is not None:
        _COMM.close()
        _COMM = None


def gather_floats(values, comm=None):
    """
    All ranks' per-character log-likelihoods in character order (shards are contiguous blocks): every rank places its
    block in a zero vector of the total length and the vectors are summed -- one small all-reduce, no all-gather needed.
    """
    comm = comm or _COMM or LocalCommunicator()
    values = np.atleast_1d(np.asarray(values, dtype=np.float64))
    counts = comm.allreduce(np.eye(comm.world)[comm.rank] * len(values)).astype(np.int64)
    out = np.zeros(int(counts.sum()))
    begin = int(counts[:comm.rank].sum())
    out[begin:begin + len(values)] = values
    return comm.allreduce(out)
