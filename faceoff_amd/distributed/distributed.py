"""Rank/world helpers and small collectives -- API of reference distributed/distributed.py:14-143.

Differences by design: `all_gather` moves Python objects with
torch.distributed.all_gather_object instead of pickling through CUDA byte tensors with two host
syncs per call (:75-107); everything is a no-op when no process group is initialised, like the
reference (:16-23,54-68).
"""
from torch import distributed as dist
from torch.utils import data

LOCAL_PROCESS_GROUP = None


def _ready():
    return dist.is_available() and dist.is_initialized()


def is_primary():
    return get_rank() == 0


def get_rank():
    return dist.get_rank() if _ready() else 0


def get_local_rank():
    if not _ready():
        return 0
    if LOCAL_PROCESS_GROUP is None:
        raise ValueError("faceoff_amd.distributed.LOCAL_PROCESS_GROUP is None")
    return dist.get_rank(group=LOCAL_PROCESS_GROUP)


def synchronize():
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def all_reduce(tensor, op=dist.ReduceOp.SUM):
    """In place; returns the tensor; no-op at world size 1 (reference :64-72)."""
    if get_world_size() == 1:
        return tensor
    dist.all_reduce(tensor, op=op)
    return tensor


def all_gather(data_obj):
    """list of every rank's object (reference :75-107)."""
    world = get_world_size()
    if world == 1:
        return [data_obj]
    out = [None] * world
    dist.all_gather_object(out, data_obj)
    return out


def reduce_dict(input_dict, average=True):
    """Reduce a dict of scalars tensors to rank 0 (reference :110-132)."""
    import torch
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        keys = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in keys], 0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return {k: v for k, v in zip(keys, values)}


def data_sampler(dataset, shuffle, distributed):
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    return data.RandomSampler(dataset) if shuffle else data.SequentialSampler(dataset)
