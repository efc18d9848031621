"""ctypes binding of the C-ABI communicator (include/faceoff_hip.h: fo_comm_*, csrc/comm.cpp): RCCL behind the same shared library as
the kernels, for a host that does not go through torch.distributed (what the reference binds at distributed/launch.py:61-66).

    comm = AbiComm.create(rank, world, device, exchange)     # exchange(id_bytes_or_None) -> id_bytes: the job's own rendezvous
    comm.allreduce_async(flat_grads[lo:hi])                   # in-place SUM, behind the current stream's work
    comm.wait()                                               # the current stream waits for every all-reduce issued so far

`exchange` is any function that hands rank 0's 128 bytes to every rank; `exchange_via_torch_store` does it over the TCP store that
`torch.distributed.init_process_group` already opened (the reference's dist_url), `exchange_single` is the one-rank case."""
import ctypes as C

import torch

from .. import _lib


def exchange_single(idb):
    return idb


def exchange_via_torch_store(group=None, key="faceoff_amd/fo_comm_id"):
    """Rank 0 publishes the id through torch.distributed's broadcast of a byte tensor on the (CPU / gloo or nccl) group."""
    from torch import distributed as dist

    def f(idb):
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if idb is not None:
            t.copy_(torch.frombuffer(bytearray(idb), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        return bytes(t.cpu().numpy().tobytes())
    return f


class AbiComm:
    def __init__(self, handle, rank, world, device):
        self._h, self.rank, self.world, self.device = handle, rank, world, torch.device(device)

    @classmethod
    def create(cls, rank, world, device, exchange=exchange_single):
        device = torch.device(device)
        idb = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _lib.call("fo_comm_unique_id", buf)
            idb = buf.raw
        idb = exchange(idb)
        assert idb is not None and len(idb) == 128
        h = C.c_void_p()
        _lib.call("fo_comm_init", C.byref(h), rank, world, C.create_string_buffer(idb, 128), device.index or 0)
        self = cls(h, rank, world, device)
        self._place_stream()
        return self

    def _place_stream(self):
        """The collectives' stream must not share the compute stream's hardware queue (HIP hands queues out at first use, four by default: on a shared queue an
        all-reduce lines up behind the kernels it is meant to overlap).  A pooled torch stream is checked with the engine's launch-overlap probe and handed to
        the communicator; the next one is tried if it fails (at most 8).  FACEOFF_NO_QUEUE_CHECK=1: the communicator keeps the stream it created."""
        import os
        if os.environ.get("FACEOFF_NO_QUEUE_CHECK") or self.device.type != "cuda":
            return
        from ..engine import _runs_beside
        with torch.cuda.device(self.device):
            for _ in range(8):
                st = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)              # (first use: the stream takes its queue)
                if _runs_beside(st, self.device):
                    self._stream = st                                # (kept alive with the communicator)
                    _lib.call("fo_comm_set_stream", self._h, C.c_void_p(st.cuda_stream))
                    return

    @property
    def issued(self):
        return int(_lib.load().fo_comm_issued(self._h))

    def allreduce_async(self, t, after_stream=None):
        """t: a contiguous fp32 CUDA tensor (a slice of the gradient arena); summed over ranks in place, behind the work already enqueued
        on after_stream (default: the current stream)."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        s = after_stream if after_stream is not None else torch.cuda.current_stream(t.device)
        _lib.call("fo_comm_allreduce_async", self._h, C.c_void_p(t.data_ptr()), C.c_int64(t.numel()), C.c_void_p(s.cuda_stream))

    def broadcast_async(self, t, root=0, after_stream=None):
        """t on every rank := rank `root`'s t (contiguous fp32 CUDA tensor), ordered like allreduce_async.  DDP's broadcast_buffers for the
        buffers that are not summed over ranks (the discriminators' InstanceNorm running statistics)."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        s = after_stream if after_stream is not None else torch.cuda.current_stream(t.device)
        _lib.call("fo_comm_broadcast_async", self._h, C.c_void_p(t.data_ptr()), C.c_int64(t.numel()), int(root), C.c_void_p(s.cuda_stream))

    def wait(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.call("fo_comm_wait", self._h, C.c_void_p(s.cuda_stream))

    def destroy(self):
        if self._h is not None:
            _lib.call("fo_comm_destroy", self._h)
            self._h = None
