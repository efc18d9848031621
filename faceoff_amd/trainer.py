"""The training iteration of train_faceoff_perceptual.py:92-107 on the MI355X engine.

    model.zero_grad(); run_step(); loss = recon + 1*latent + 1*perceptual; loss.backward();
    scheduler.step(); optimizer.step()

becomes: engine forward -> fused loss kernels -> engine backward (gradient buckets all-reduced on a
side stream as they complete) -> one multi-tensor Adam launch over the flat arena.  No host
synchronisation anywhere in the step: losses stay on the device until the caller asks.
"""
from __future__ import annotations

import os as _os

import torch

from . import ops
from .distributed import GradBucketReducer, fused_vq_allreduce, get_world_size
from .engine import VQVAEEngine

LATENT_LOSS_WEIGHT = 1.0       # reference config.py:5
PERCEPTUAL_LOSS_WEIGHT = 1.0   # reference config.py:6


class FlatAdam:
    """torch.optim.Adam(model.parameters(), lr) (train_faceoff_perceptual.py:190) as ONE launch."""

    def __init__(self, engine: VQVAEEngine, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
        self.engine = engine
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps)]
        self.m = torch.zeros_like(engine.flat_params)
        self.v = torch.zeros_like(engine.flat_params)
        self.t = 0

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.t += 1
        ops.adam_flat(self.engine.flat_params, self.engine.flat_grads, self.m, self.v, g["lr"], self.t, g["betas"],
                      g["eps"], grad_scale)
        self.engine.mark_params_dirty()          # (the launch writes the arena through a raw pointer: torch's version counter does not see it)

    def state_dict(self):
        return dict(m=self.m, v=self.v, t=self.t, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.t = sd["t"]; self.param_groups = sd["param_groups"]


class FaceOffTrainer:
    def __init__(self, engine: VQVAEEngine, lr=3e-4, scheduler=None, vqlpips=None, bucket_bytes=4 << 20, group=None,
                 force_collectives=False, comm=None):
        """force_collectives: run the gradient-bucket and VQ-statistics all-reduces even in a one-rank process group
        (tests: the RCCL path on a single GPU).  comm: a distributed.comm.AbiComm -- gradients and VQ statistics then travel through
        the C-ABI communicator (fo_comm_*) instead of torch.distributed."""
        self.engine = engine
        self.optimizer = FlatAdam(engine, lr=lr)
        self.scheduler = scheduler
        self.vqlpips = vqlpips
        self.world = comm.world if comm is not None else get_world_size()
        # the ground-truth branch of LPIPS does not depend on the model: it runs on its own stream beside the VQ-VAE
        # forward (an fp32 conv workgroup and a bf16 one fit a CU's LDS together; the bf16 convs are L2-bound,
        # the fp32 ones matrix-pipe-bound)
        import os as _os
        self.lpips_stream = None
        if vqlpips is not None and engine.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_LPIPS_OVERLAP"):
            self.lpips_stream = torch.cuda.Stream(device=engine.device)
        # Flow control: the host may enqueue at most this many steps ahead of the GPU.  A training loop that never reads a loss lets the host
        # run arbitrarily far ahead, and with several streams per step the caching allocator then cannot hand a step's blocks to the next
        # one (their uses are still pending): the pool grew by 20-60 GB a few steps into a run -- device mallocs of gigabytes in the middle
        # of training (tools/probes/c3_steps_probe.py).  Two steps in flight keep the GPU fed; the wait costs nothing when it is the bottleneck.
        self.max_inflight_steps = int(_os.environ.get("FACEOFF_MAX_INFLIGHT_STEPS", "2"))
        self.claim_queues()
        self._inflight = []
        self.reducer = None
        if self.world > 1 or force_collectives:
            self.reducer = GradBucketReducer(engine.flat_grads, engine.layer_order, engine.offsets, bucket_bytes, group,
                                             always=force_collectives, comm=comm)
            engine.grad_ready_hook = self.reducer.layer_done
            if comm is not None:
                def vq_ar_abi(st):             # in the forward's critical path: summed behind the current stream, awaited by it
                    comm.allreduce_async(st)
                    comm.wait()
                    return st
                engine.vq_allreduce = vq_ar_abi
            else:
                vq_ar = fused_vq_allreduce(group)
                engine.vq_allreduce = (lambda st: vq_ar(st, always=True)) if force_collectives else vq_ar

    def claim_queues(self):
        """The side streams get their hardware queues by plan (see VQVAEEngine.keep_wgrad_off_main_queue for why the order of first use is not to be relied on: the
        queues HIP hands out depend on every stream the process used before).  The plan is the mapping the step was tuned on -- what a fresh process gets by reaching
        the streams in the order of a step: LPIPS' ground-truth branch alone, filter packs + filter gradients (never busy together), bottom Conv3d chain + quantiser
        statistics, nothing on the main stream's queue -- made explicit with engine.streams_by_queue, so that it also holds behind a communicator's or another
        engine's streams: config 3 31.85-31.93 ms with two or four foreign streams in front where the order of first use gave 33.0-33.6, config 2 39.8-39.9 where it
        gave 40.2-41.5; in a fresh process the same as before (32.0 / 39.9).  ~50 ms and a transient 512 MB.  FACEOFF_NO_QUEUE_PLAN=1: order of first use, with the
        filter-gradient stream checked."""
        eng = self.engine
        if eng.device.type != "cuda":
            return
        plan = None
        eng.diag_queue_shift()
        if not _os.environ.get("FACEOFF_NO_QUEUE_PLAN") and not _os.environ.get("FACEOFF_NO_QUEUE_CHECK"):
            from .engine import streams_by_queue
            plan = streams_by_queue(eng.device)
        if plan is None:
            if self.lpips_stream is not None:                  # (first use, in the order a step reaches the streams: the ground-truth branch comes first)
                with torch.cuda.stream(self.lpips_stream):
                    torch.zeros(1, device=eng.device)
            eng.keep_wgrad_off_main_queue()
            return
        (a, b, c), _ = plan
        if self.lpips_stream is not None:
            self.lpips_stream = a[0]
        else:                                                  # (no perceptual branch: the packs take the free queue)
            b = [a[0], b[0]]
        if eng.pack_stream is not None:
            eng.pack_stream = b[0]
        if eng.wgrad_stream is not None:
            eng.wgrad_stream = b[1] if self.lpips_stream is not None else c[1]
        if eng.aux_stream is not None:
            eng.aux_stream = c[0] if self.lpips_stream is not None else b[1]
        if eng.vq_stream is not None:
            eng.vq_stream = c[1] if self.lpips_stream is not None else c[0]
        eng._streams = (eng.wgrad_stream, eng.aux_stream, eng.pack_stream, eng.vq_stream)       # (set_stream_overlap(True) restores from this tuple)

    def step(self, img, ground_truth, T=None, force_ids=None):
        """img [B,T,6,H,W] or [N,6,H,W]; ground_truth likewise with 3 channels (utils.py:29-38).
        Returns device scalars (recon_loss, latent_loss, perceptual_loss).  force_ids: teacher-forced codes (VQVAEEngine.forward)."""
        if isinstance(img, (tuple, list)):       # (source, background): concatenated inside the input-layout kernel
            if img[0].dim() == 5:
                T = T or img[0].shape[1]
                img = tuple(t.reshape(-1, *t.shape[2:]) for t in img)
                ground_truth = ground_truth.reshape(-1, *ground_truth.shape[2:])
        elif img.dim() == 5:
            T = T or img.shape[1]
            img = img.reshape(-1, *img.shape[2:])
            ground_truth = ground_truth.reshape(-1, *ground_truth.shape[2:])
        eng = self.engine
        if eng.device.type == "cuda" and self.max_inflight_steps > 0:
            while len(self._inflight) >= self.max_inflight_steps:
                self._inflight.pop(0).synchronize()
        taps0 = None
        if self.vqlpips is not None and self.lpips_stream is not None and eng.wgrad_stream is not None:
            main = torch.cuda.current_stream()
            self.lpips_stream.wait_stream(main)
            ground_truth.record_stream(self.lpips_stream)
            with torch.cuda.stream(self.lpips_stream):
                taps0 = self.vqlpips.target_taps(ground_truth)
            if taps0 is not None:
                for t in taps0:
                    t.record_stream(main)        # allocated on the side stream, read (and freed) on the main one
        S = eng.forward(img, training=True, T=T, force_ids=force_ids)
        self.last_ids = (S["id_t"], S["id_b"])           # the codes of the step just enqueued (two small int64 tensors)
        if getattr(self, "keep_states", False):          # parity tests only: the forward state (9.6 GB at C2) stays alive until the next step
            self.last_state = S                          # (its ReLU branches are read back: tests/_fullsize_oracle.py engine_relu_masks)
        dec = S["dec"]
        acc = torch.zeros(1, device=eng.device)
        one = torch.ones(1, device=eng.device)
        g_dec = torch.empty_like(dec)
        ops.mse_slice_fwd_bwd(dec, ground_truth, acc, one, g_dec)       # loss value and gradient in one pass over dec and gt
        recon = acc / float(ground_truth.numel())
        perceptual = torch.zeros(1, device=eng.device)
        if self.vqlpips is not None:
            if taps0 is not None:
                torch.cuda.current_stream().wait_stream(self.lpips_stream)
            self.vqlpips._bind(dec.device).head_overlap = eng.wgrad_stream is not None      # side streams folded (set_stream_overlap(False)): the heads too
            perceptual = self.vqlpips.loss_and_grad(ground_truth, dec, g_dec, PERCEPTUAL_LOSS_WEIGHT, taps0=taps0)
        eng.backward(S, g_dec, one * LATENT_LOSS_WEIGHT)
        if self.reducer is not None:
            self.reducer.finish()
        if self.scheduler is not None:           # before optimizer.step(), as the reference (:104-107)
            self.scheduler.step()
        self.optimizer.step(grad_scale=1.0 / self.world)   # DDP averages gradients
        if eng.device.type == "cuda" and self.max_inflight_steps > 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(eng.device))
            self._inflight.append(ev)
        return recon, S["diff"], perceptual

    def run_host_fed(self, loader, max_steps=None):
        """The training loop body over a loader of CPU 5-tuples with the host -> HBM copies double-buffered on a copy stream
        (faceoff_amd.feeder.HostFedBatches): yields (recon, latent, perceptual, T) device scalars per step."""
        from .feeder import HostFedBatches
        for i, (parts, T, ground_truth) in enumerate(HostFedBatches(loader, self.engine.device)):
            if max_steps is not None and i >= max_steps:
                break
            recon, latent, perceptual = self.step(parts, ground_truth, T=T)
            yield recon, latent, perceptual, T

    def step_from_batch(self, data):
        """One iteration from the loader's 5-tuple (reference train loop :95 `process_data` + run_step + backward +
        optimizer): returns (recon_loss, latent_loss, perceptual_loss, S) like run_step (:32-47), losses as device scalars."""
        from .utils import split_batch
        parts, T, ground_truth = split_batch(data, self.engine.device)
        recon, latent, perceptual = self.step(parts, ground_truth, T=T)
        return recon, latent, perceptual, T
