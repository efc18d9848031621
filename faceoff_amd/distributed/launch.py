"""One process per GPU.  `launch(fn, n_gpu_per_machine, n_machine, machine_rank, dist_url, args)` has the reference's
signature and meaning (distributed/launch.py:22; called as `dist.launch(main, args.n_gpu, 1, 0, args.dist_url,
args=(args,))`, train_faceoff_perceptual.py:253): with a world of one `fn(*args)` runs in the calling process, otherwise
one fresh interpreter per local GPU joins a `torch.distributed` group -- backend "nccl" (= RCCL over xGMI on ROCm) when
the machine has GPUs, "gloo" otherwise (CPU tests) -- gets the per-machine subgroup behind `get_local_rank()`, and runs
`fn(*args)`.

Children are started with the "spawn" method: a child never inherits an initialised HIP runtime and nothing re-execs
after touching the GPU.  A failing rank fails the launch (the others are terminated), like mp.spawn in the reference.
"""
import os
import socket
from dataclasses import dataclass

import torch
from torch import distributed as dist
from torch import multiprocessing as mp

from . import distributed as dist_fn


@dataclass
class _Rank:
    local: int
    per_machine: int
    machine: int
    world: int
    url: str
    backend: str

    @property
    def index(self):
        return self.machine * self.per_machine + self.local


def find_free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rendezvous_url(dist_url, n_machine):
    if dist_url in (None, "auto"):
        if n_machine > 1:
            raise ValueError("an automatic rendezvous address only works on one machine: pass dist_url='tcp://host:port'")
        return f"tcp://127.0.0.1:{find_free_port()}"
    if n_machine > 1 and dist_url.startswith("file://"):
        raise ValueError("file:// rendezvous needs a shared file system and is unreliable across machines: use tcp://")
    return dist_url


def launch(fn, n_gpu_per_machine, n_machine=1, machine_rank=0, dist_url=None, args=(), backend=None):
    world = n_machine * n_gpu_per_machine
    if world <= 1:
        return fn(*args)
    if backend is None:
        backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"     # device_count() does not initialise HIP
    if backend == "nccl" and torch.cuda.device_count() < n_gpu_per_machine:
        raise ValueError(f"{n_gpu_per_machine} ranks per machine requested, {torch.cuda.device_count()} GPUs present")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's driver supports dmabuf IPC only
    url = _rendezvous_url(dist_url, n_machine)
    ctx = mp.get_context("spawn")
    procs = []
    for local in range(n_gpu_per_machine):
        spec = _Rank(local, n_gpu_per_machine, machine_rank, world, url, backend)
        p = ctx.Process(target=_rank_main, args=(spec, fn, args), daemon=False)
        p.start()
        procs.append(p)
    failed = None
    while failed is None and any(p.is_alive() for p in procs):
        for p in procs:
            p.join(timeout=0.2)
            if p.exitcode not in (None, 0):
                failed = p
                break
    if failed is None:
        failed = next((p for p in procs if p.exitcode != 0), None)
    if failed is not None:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join()
        raise RuntimeError(f"rank process {procs.index(failed)} exited with code {failed.exitcode}")


def _rank_main(spec, fn, args):
    kw = {}
    if spec.backend == "nccl":
        torch.cuda.set_device(spec.local)
        kw["device_id"] = torch.device("cuda", spec.local)
    dist.init_process_group(backend=spec.backend, init_method=spec.url, world_size=spec.world, rank=spec.index, **kw)
    try:
        dist_fn.synchronize()
        # every rank creates every machine's subgroup (new_group is collective); it keeps its own
        for m in range(spec.world // spec.per_machine):
            group = dist.new_group(list(range(m * spec.per_machine, (m + 1) * spec.per_machine)))
            if m == spec.machine:
                dist_fn.LOCAL_PROCESS_GROUP = group
        fn(*args)
    finally:
        dist.destroy_process_group()
