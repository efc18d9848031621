"""Mirror of the reference's `distributed` package (distributed/__init__.py:1-13) over RCCL.

Same names, arguments and single-process degradation as the reference helpers; `launch` starts one
process per GPU with torch.distributed backend "nccl" (= RCCL over xGMI on ROCm).  Added for the
MI355X engine: `GradBucketReducer` (bucketed, backward-overlapped gradient all-reduce) and
`fused_vq_allreduce` (one message instead of the reference's two blocking ones per quantiser).
"""
from .distributed import (
    get_rank,
    get_local_rank,
    is_primary,
    synchronize,
    get_world_size,
    all_reduce,
    all_gather,
    reduce_dict,
    data_sampler,
    LOCAL_PROCESS_GROUP,
)
from .launch import launch
from .reducer import GradBucketReducer, fused_vq_allreduce

__all__ = ["get_rank", "get_local_rank", "is_primary", "synchronize", "get_world_size", "all_reduce", "all_gather",
           "reduce_dict", "data_sampler", "LOCAL_PROCESS_GROUP", "launch", "GradBucketReducer", "fused_vq_allreduce"]
