"""One process per GPU -- API of reference distributed/launch.py:22-92.

`launch(fn, n_gpu_per_machine, n_machine, machine_rank, dist_url, args)` keeps the reference
signature.  Backend is "nccl" (RCCL) when GPUs are present, "gloo" otherwise (CPU tests).  The
child must never exec after touching the GPU; mp.spawn starts fresh interpreters, which is safe.
"""
import os

import torch
from torch import distributed as dist
from torch import multiprocessing as mp

from . import distributed as dist_fn


def find_free_port():
    import socket
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch(fn, n_gpu_per_machine, n_machine=1, machine_rank=0, dist_url=None, args=(), backend=None):
    world_size = n_machine * n_gpu_per_machine
    if world_size <= 1:
        return fn(*args)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist_url in (None, "auto"):
        if n_machine != 1:
            raise ValueError('dist_url="auto" not supported in multi-machine jobs')
        dist_url = f"tcp://127.0.0.1:{find_free_port()}"
    if n_machine > 1 and dist_url.startswith("file://"):
        raise ValueError("file:// is not a reliable init method in multi-machine jobs. Prefer tcp://")
    mp.spawn(distributed_worker, nprocs=n_gpu_per_machine,
             args=(fn, world_size, n_gpu_per_machine, machine_rank, dist_url, args, backend), daemon=False)


def distributed_worker(local_rank, fn, world_size, n_gpu_per_machine, machine_rank, dist_url, args, backend=None):
    if backend is None:
        backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"
    if backend == "nccl":
        if n_gpu_per_machine > torch.cuda.device_count():
            raise ValueError(f"specified n_gpu_per_machine larger than available device ({torch.cuda.device_count()})")
        torch.cuda.set_device(local_rank)
    global_rank = machine_rank * n_gpu_per_machine + local_rank
    try:
        dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=global_rank)
    except Exception as e:
        raise OSError(f"failed to initialize {backend} groups: {e}")
    dist_fn.synchronize()
    if dist_fn.LOCAL_PROCESS_GROUP is not None:
        raise ValueError("faceoff_amd.distributed.LOCAL_PROCESS_GROUP is not None")
    n_machine = world_size // n_gpu_per_machine
    for i in range(n_machine):
        ranks_on_i = list(range(i * n_gpu_per_machine, (i + 1) * n_gpu_per_machine))
        pg = dist.new_group(ranks_on_i)
        if i == machine_rank:
            dist_fn.LOCAL_PROCESS_GROUP = pg
    try:
        fn(*args)
    finally:
        dist.destroy_process_group()
