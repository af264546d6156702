"""
Multi-GPU layout of the path: characters (columns) are independent (pastml/acr.py:213-231 runs one ml_acr per
character), so they are sharded over the ranks of one node -- one process per GPU -- with the tree replicated.
No collective touches the data path; the only exchange is the (all-)reduce of the summed log-likelihood, an 8-byte
message over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).
"""
import os


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


def allreduce_sum(value, device=None, group=None):
    """
    Sum of a python float over the ranks (identity without an initialised process group).  Deterministic for a fixed
    world size: a single scalar per rank.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


def gather_floats(values, device=None, group=None):
    """All ranks' per-character log-likelihoods (equal shard sizes), in character order: list of lists -> flat list."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device if device is not None else 'cpu')
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [float(v) for o in out for v in o.tolist()]
