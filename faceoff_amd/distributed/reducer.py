"""Data-parallel gradient exchange for the flat gradient arena (replaces nn.parallel.
DistributedDataParallel at train_faceoff_perceptual.py:164-169).

The engine's backward fills the arena from its END towards its START (VQVAEEngine.layer_order), so a
bucket is a contiguous slice that becomes final when the layer at its low end is done.  As soon as
that happens the slice is all-reduced (SUM; averaging is folded into the optimiser's grad_scale) on
a side HIP stream, overlapping the rest of backward.  xGMI is point-to-point and the whole payload
is 16.2 MB, so buckets are few and large (default 4 MiB -> 4-5 messages): latency, not bandwidth,
is what there is to hide.
"""
import torch
from torch import distributed as dist


class GradBucketReducer:
    def __init__(self, flat_grads, layer_order, offsets, bucket_bytes=4 << 20, group=None, always=False, comm=None):
        """layer_order: arena order of layer names (reverse of backward completion);
        offsets: key -> (offset, numel) for '<layer>.weight' / '<layer>.bias'.
        comm: a distributed.comm.AbiComm -- the buckets then go through the C-ABI communicator (fo_comm_allreduce_async on ITS stream,
        fo_comm_wait) instead of torch.distributed."""
        self.flat = flat_grads
        self.group = group
        self.comm = comm
        if comm is not None:
            self.world = comm.world
            self.active = self.world > 1 or always
        else:
            self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
            # always=True: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU)
            self.active = self.world > 1 or (always and dist.is_available() and dist.is_initialized())
        # walk the arena from the end (first-completed layer) building buckets
        self.buckets = []          # (lo, hi, last layer of the bucket in backward order)
        self.members = []          # layer names per bucket
        hi = flat_grads.numel()
        cur_hi, cur = hi, []
        for name in reversed(layer_order):
            lo = offsets[name + ".weight"][0]
            cur.append(name)
            if (cur_hi - lo) * 4 >= bucket_bytes:
                self.buckets.append((lo, cur_hi, name))
                self.members.append(cur)
                cur_hi, cur = lo, []
        if cur_hi > 0:
            self.buckets.append((0, cur_hi, layer_order[0]))
            self.members.append(cur)
        # a bucket fires when ALL its layers have reported (the engine overlaps independent chains, so layers
        # do not finish strictly in arena order).  Members may report from DIFFERENT streams (filter-gradient side
        # stream, the bottom Conv3d chain's stream, or the main stream when an overlap switch is off), so each member
        # records its own event on the stream that is current when it reports, and the all-reduce waits on all of them.
        self._bucket_of = {n: i for i, ms in enumerate(self.members) for n in ms}
        self._pending = [set(ms) for ms in self.members]
        self._events = [[] for _ in self.members]
        self._works = []
        self.cuda = flat_grads.is_cuda
        self.side = torch.cuda.Stream(device=flat_grads.device) if self.cuda and self.active else None
        self.launched = []         # bucket indices in launch order (tests)
        # bench.py's `comm` block: with time_exposed = True, finish() brackets the main stream's wait for the side stream with HIP
        # events; their distance is the all-reduce time that backward did NOT cover (from "compute stream reached the join" to "last
        # bucket summed")
        self.time_exposed = False
        self._exposed = []

    def layer_done(self, name):
        """grad_ready_hook of the engine: fire the bucket whose last layer just completed."""
        i = self._bucket_of.get(name)
        if i is None or not self.active:
            return
        self._pending[i].discard(name)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._events[i].append(ev)
        if self._pending[i]:
            return
        lo, hi, _ = self.buckets[i]
        self.launched.append(i)
        view = self.flat[lo:hi]
        if self.cuda:
            for ev in self._events[i]:
                self.side.wait_event(ev)
            self._events[i] = []
            if self.comm is not None:          # the communicator's own stream runs the collective, behind the side stream's waits
                self.comm.allreduce_async(view, after_stream=self.side)
                return
            with torch.cuda.stream(self.side):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Join the side stream; afterwards the arena holds the SUM over ranks."""
        if not self.active:
            return
        if self.cuda:
            if self.comm is not None:
                self.comm.wait(self.side)      # (the join below then covers the collectives)
            with torch.cuda.stream(self.side):
                for w in self._works:
                    w.wait()
            main = torch.cuda.current_stream()
            if self.time_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                main.wait_stream(self.side)
                e1.record(main)
                self._exposed.append((e0, e1))
            else:
                main.wait_stream(self.side)
        else:
            for w in self._works:
                w.wait()
        self._works = []
        self.launched = []
        self._pending = [set(ms) for ms in self.members]
        self._events = [[] for _ in self.members]


    def exposed_ms(self, reset=True):
        """Mean per step of the time the compute stream spent waiting for the gradient all-reduces (synchronises); None if
        nothing was timed."""
        if not self._exposed:
            return None
        torch.cuda.synchronize(self.flat.device)
        ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        if reset:
            self._exposed = []
        return ms

    def describe(self):
        """Static facts for bench.py's `comm` block."""
        return {"buckets": len(self.buckets), "bucket_bytes": [(hi - lo) * 4 for lo, hi, _ in self.buckets],
                "grad_bytes_per_step": int(self.flat.numel()) * 4}


def fused_vq_allreduce(group=None):
    """Returns f(stats[512 + 512*64]) summing the EMA statistics of one quantiser over ranks in ONE
    message (the reference issues two blocking all-reduces per quantiser, vqvae_conv3d_latent.py:63-64)."""
    def f(stats, always=False):
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        return stats
    return f
