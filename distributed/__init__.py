"""The reference's top-level `distributed` package (distributed/__init__.py:1-13; `import distributed as dist` in the
trainer, `import distributed as dist_fn` in models/vqvae_conv3d_latent.py:7) over RCCL: the same eleven names."""
from faceoff_amd.distributed import (  # noqa: F401
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
    launch,
)
from faceoff_amd.distributed import distributed, launch as _launch_mod  # noqa: F401
