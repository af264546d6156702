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
* otherwise                        -> :class:`RcclCommunicator`; if it cannot be set up (no librccl, ...) a warning and
                                      torch.distributed's nccl backend behind the same interface (PASTML_AMD_COMM=torch-nccl
                                      asks for it directly)
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
        self._eng = engine if engine is not None else hip.BareContext(device)
        forced = bool(os.environ.get('PASTML_HIP_COMM_FORCE_RCCL'))
        uid, path = exchange_unique_id(rank, hip.comm_unique_id) if (world > 1 or forced) else (None, None)
        self._eng.comm_init(rank, world, uid)   # collective: returns once every rank has read the id
        if rank == 0 and path:
            try:
                os.remove(path)
                os.rmdir(os.path.dirname(path))
            except OSError:
                pass

    def allreduce(self, values, op='sum'):
        from pastml_amd import hip
        return self._eng.allreduce(values, hip.COMM_SUM if op == 'sum' else hip.COMM_MAX)

    def allreduce_loglik(self, loglik):
        return self._eng.allreduce_loglik(loglik)

    def barrier(self):
        self._eng.allreduce([0.0])

    def close(self):
        if self._eng is not None:
            self._eng.comm_destroy()
            if self._own:
                self._eng.close()
            self._eng = None


class TorchCommunicator(object):
    """
    torch.distributed behind the same interface.  'gloo' (CPU): tests / dry runs of the N > 1 path on boxes with fewer
    GPUs than ranks.  'nccl' (= RCCL through torch): the fallback if the library's own communicator cannot be set up.
    """

    def __init__(self, rank, world, backend='gloo', device=None):
        import torch
        import torch.distributed as dist
        self._dist = dist
        self.name = backend if backend == 'gloo' else 'torch-' + backend
        self._device = 'cpu'
        self._own = not dist.is_initialized()
        if backend == 'nccl':
            local = rank_world()[2] if device is None else device
            torch.cuda.set_device(local)
            self._device = 'cuda:{}'.format(local)
            if self._own:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(self._device))
        elif self._own:
            dist.init_process_group(backend, rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def allreduce(self, values, op='sum'):
        import torch
        t = torch.tensor(np.atleast_1d(values), dtype=torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == 'sum' else self._dist.ReduceOp.MAX)
        return t.cpu().numpy().copy()

    def allreduce_loglik(self, loglik):
        return float(self.allreduce([LocalCommunicator().allreduce_loglik(loglik)])[0])

    def barrier(self):
        self.allreduce([0.0])

    def close(self):
        if self._own and self._dist.is_initialized():
            self._dist.destroy_process_group()


def GlooCommunicator(rank, world):
    return TorchCommunicator(rank, world, 'gloo')


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
    elif kind == 'torch-nccl':
        _COMM = TorchCommunicator(rank, world, 'nccl', device=device)
    else:
        try:
            _COMM = RcclCommunicator(rank, world, device=local_rank if device is None else device, engine=engine)
        except Exception as e:   # librccl missing, no id from rank 0, ...: the same failure on every rank of the node
            import sys
            sys.stderr.write('pastml_amd.sharding: the library\'s RCCL communicator failed ({}); falling back to '
                             'torch.distributed (nccl)\n'.format(e))
            _COMM = TorchCommunicator(rank, world, 'nccl', device=device)
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
