"""Host-fed input path (SURVEY section 8 f4 "feeding"; reference utils.py:29-38,69-73).

The reference moves one clip per step to the GPU inside `process_data` (`.to(device)` of three [T,3,H,W] tensors plus the unused
`source_images_original`), synchronously, on the compute stream.  At BASELINE config 2 that is 377 MB per step (160 frames x 9 channels
x 256 x 256 x 4 B): 6 ms over PCIe Gen5 if it is exposed.  Here the loader's batches go through

    pinned host buffers  --(copy stream, hipMemcpyAsync)-->  one of two device buffer sets  --(event)-->  the compute stream,

so that the copy of batch n+1 runs beside the step on batch n; `source_images_original` is never moved (training does not read it).
The device tensors handed out are exactly what the resident path would hold, so a fed step is bit-identical to a resident one
(tests/test_feeder_gpu.py).
"""
from __future__ import annotations

import torch


def _frames(x):
    return x.reshape(-1, *x.shape[-3:]) if x.dim() == 5 else x


class HostFedBatches:
    """Iterate over `loader` (an iterable of the reference's 5-tuples of CPU tensors, each [1,T,3,H,W] or [B,T,3,H,W]) yielding
    ((source, background), T, ground_truth) on `device`, double-buffered: while the consumer's step runs on batch n, batch n+1 is already
    being copied on a side stream.  Pinned staging buffers are allocated once (a loader built with pin_memory=True skips the staging copy)."""

    def __init__(self, loader, device, depth=2):
        if int(depth) < 2:
            raise ValueError(f"faceoff_amd: HostFedBatches needs depth >= 2 (one slot being read by the step, one being filled), got {depth}")
        self.loader = loader
        self.device = torch.device(device)
        self.depth = int(depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._dev = [None] * depth         # per slot: (source, background, ground_truth) device tensors
        self._pin = [None] * depth         # per slot: pinned host staging tensors
        self._ready = [None] * depth       # per slot: event recorded on the copy stream after the slot's copies
        self._free = [None] * depth        # per slot: event recorded on the consumer's stream when it has finished with the slot

    def _stage(self, slot, data):
        source, target, background, source_images, _ = data                # (source_images_original stays on the host: SURVEY a0)
        T = source.shape[-4]
        host = [_frames(t) for t in (source, background, source_images)]
        for h in host:
            if h.dtype != torch.float32 or h.is_cuda:
                raise TypeError(f"faceoff_amd: HostFedBatches takes the loader's float32 CPU tensors (ToTensor + Normalize output), got {h.dtype} on {h.device}")
        if self._dev[slot] is None or any(d.shape != h.shape for d, h in zip(self._dev[slot], host)):
            self._dev[slot] = [torch.empty(h.shape, dtype=torch.float32, device=self.device) for h in host]
            self._pin[slot] = [None if h.is_pinned() else torch.empty(h.shape, dtype=torch.float32).pin_memory() for h in host]
        if self._ready[slot] is not None:
            self._ready[slot].synchronize()                                # the slot's previous copies have left the pinned staging buffers
        if self._free[slot] is not None:
            self.copy_stream.wait_event(self._free[slot])                  # the step that read this slot's device tensors is done
        with torch.cuda.stream(self.copy_stream):
            for h, p, d in zip(host, self._pin[slot], self._dev[slot]):
                src = h if p is None else p.copy_(h)                       # (pageable -> pinned staging on the host thread)
                d.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[slot] = ev
        return T

    def __iter__(self):
        it = iter(self.loader)
        pending = []                                                       # (slot, T) of batches whose copies are in flight
        slot = 0
        try:
            for _ in range(self.depth - 1):
                pending.append((slot, self._stage(slot, next(it))))
                slot = (slot + 1) % self.depth
        except StopIteration:
            pass
        while pending:
            try:                                                           # keep depth - 1 batches in flight behind the one handed out
                pending.append((slot, self._stage(slot, next(it))))
                slot = (slot + 1) % self.depth
            except StopIteration:
                pass
            cur, T = pending.pop(0)
            torch.cuda.current_stream(self.device).wait_event(self._ready[cur])
            src, bg, gt = self._dev[cur]
            yield (src, bg), T, gt
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))              # the consumer's launches that read the slot are all enqueued
            self._free[cur] = ev
