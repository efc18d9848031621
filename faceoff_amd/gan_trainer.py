"""The two-optimiser GAN iteration of the reference's disc_trainers/train_vqvae_mocoganhd_disc.py (:303-432, BASELINE
config 5) on the MI355X engines: the VQ-VAE generator (VQVAEEngine) against a video discriminator (ModelD_3d) and an image
discriminator (ModelD_img), relativistic average LSGAN, generator and discriminators updated on alternating iterations.

    even iteration (:346-383)  G_loss = recon + latent + G_loss_2d + G_loss_3d  -> generator backward -> scheduler -> Adam(G)
    odd iteration  (:385-432)  D_loss_3d -> Adam(D_3d, betas (0.5, .999));  D_loss (image) -> Adam(D_img, betas (0.5, .999))

Every iteration first runs the generator on the clip in training mode (run_step :43-53; the EMA codebooks move on
discriminator iterations too, as in the reference), cuts a window of `window` consecutive frames at a random offset (:332-335),
pairs frame 0 of the window with every later frame on the channel axis for the video discriminator (:364-365,395-396, randomly
reversed in time by flip_video :169-174) and with ONE random later frame for the image discriminator (:354-357,412-413).

Not reproduced from the reference (documented in oracle/disc_oracle.py too): the generator branch repeats the first REAL frame
window-2 instead of window-1 times (:364, a shape error at run time); `modelD.module.optim` presumes a DDP wrap.
Random choices are drawn from a `random.Random` in the reference's call order, or passed in explicitly (tests)."""
from __future__ import annotations

import os as _os
import random as _random

import torch

from . import ops
from .distributed import fused_vq_allreduce, get_world_size
from .disc import DiscEngine, make_pairs, pairs_backward, ralsgan_pair
from .engine import VQVAEEngine
from .trainer import FlatAdam, LATENT_LOSS_WEIGHT


class GANTrainer:
    def __init__(self, engine: VQVAEEngine, disc3d: DiscEngine, disc2d: DiscEngine, lr=3e-4, d_lr=1e-4, scheduler=None, window=16,
                 rng=None, comm=None, force_collectives=False):
        """comm: a distributed.comm.AbiComm -- the three gradient arenas, the VQ statistics and the discriminators' running statistics then
        travel through the C-ABI communicator (fo_comm_*) instead of torch.distributed.  force_collectives: issue every collective even in a
        one-rank world (tests: the RCCL / fo_comm path on a single GPU)."""
        self.engine, self.d3, self.d2 = engine, disc3d, disc2d
        self.comm = comm
        self.optimizer = FlatAdam(engine, lr=lr)
        self.scheduler = scheduler
        self.d_lr = d_lr
        self.window = window
        self.rng = rng if rng is not None else _random.Random()
        self.iteration = 0
        # the image discriminator (two samples of ONE frame pair: launches of a few hundred workgroups) runs on its own stream beside the video
        # discriminator -- the two are independent until their input gradients meet in the decoder-output gradient (FACEOFF_NO_D2_OVERLAP=1: serial)
        self.d2_stream = None
        if engine.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_D2_OVERLAP"):
            self.d2_stream = torch.cuda.Stream(device=engine.device)
        self.overlap_d2 = True                     # (bench.py's per-kernel region folds it, like the engine's side streams)
        # data parallel (the reference wraps generator and both discriminators in DDP): every rank runs its own clip, the
        # flat gradient arenas are summed over ranks in one all-reduce each and averaged inside the Adam launch; the VQ
        # statistics are summed in the forward (vqvae_conv3d_latent.py:63-64)
        self.world = comm.world if comm is not None else get_world_size()
        self.collectives = self.world > 1 or force_collectives
        self.collectives_issued = 0                # gradient-arena all-reduces + running-statistics broadcasts (the VQ statistics count in the engine's hook)
        if comm is not None and self.collectives:
            def vq_ar_abi(st):
                comm.allreduce_async(st)
                comm.wait()
                return st
            engine.vq_allreduce = vq_ar_abi
        elif self.collectives:
            vq_ar = fused_vq_allreduce()
            engine.vq_allreduce = (lambda st: vq_ar(st, always=True)) if force_collectives else vq_ar
        self.claim_queues()

    def claim_queues(self):
        """HIP hands a stream its hardware queue at first use, from a pool of GPU_MAX_HW_QUEUES = 4 (queue 1 is the null stream's; then 2, 3, 4, 4, 3, 2, 1, 4, 3 ..
        in order of first use: tools/stream_timeline.py), and torch hands out pooled streams that keep theirs.  This iteration uses seven side streams: left to the
        order in which the code reaches them -- and to whatever the process used before -- the generator's filter-gradient stream landed on the MAIN stream's queue
        (its launches queue up behind the data-gradient chain they are meant to run beside: 13.6 ms per iteration) or two streams that are busy together shared
        one.  The queues are therefore handed out explicitly: pooled streams are drawn and sorted by queue (engine.streams_by_queue tells queues apart by whether a
        launch on one stream waits for a long launch on the other), and the roles paired so that streams sharing a queue are never busy together --
        packs + the image discriminator's coarse scales + the quantiser statistics, bottom Conv3d chain + the video discriminator's coarse scales, filter
        gradients + image discriminator; nothing on the main stream's queue (the statistics there: 13.2).  13.1 ms in any process; ~50 ms and a transient
        512 MB at construction.  FACEOFF_NO_QUEUE_PLAN=1: streams as they come (the filter-gradient stream is still checked)."""
        eng = self.engine
        if eng.device.type != "cuda":
            return
        plan = None
        if (not _os.environ.get("FACEOFF_NO_QUEUE_PLAN") and self.d2_stream is not None and None not in (eng.pack_stream, eng.aux_stream, eng.wgrad_stream, eng.vq_stream)
                and self.d3._side() is not None and self.d2._side() is not None):
            from .engine import streams_by_queue
            plan = streams_by_queue(eng.device)
        if plan is None:
            eng.keep_wgrad_off_main_queue()
            return
        (a, b, c), _ = plan
        eng.pack_stream, self.d2._scale_stream, eng.vq_stream = a
        eng.aux_stream, self.d3._scale_stream = b[:2]
        eng.wgrad_stream, self.d2_stream = c[:2]
        eng._streams = (eng.wgrad_stream, eng.aux_stream, eng.pack_stream, eng.vq_stream)       # (set_stream_overlap(True) restores from this tuple)

    def _beside(self):
        """Context manager: the body runs on the image discriminator's side stream behind everything enqueued so far on the current stream
        (or inline when there is no side stream / overlap is off); the value it yields, called later on the main stream, joins the side stream."""
        import contextlib
        side = self.d2_stream if self.overlap_d2 else None

        @contextlib.contextmanager
        def cm():
            if side is None:
                yield (lambda: None)
                return
            # Lifetime rule for everything that crosses: tensors made inside the block live in the side stream's allocator pool and are read on
            # the main stream only after joined(); tensors of the main stream read inside were made before side.wait_stream(main).  No
            # record_stream is needed while BOTH edges hold -- so the join is not left to the caller alone: if the body raises (or a caller
            # returns early without calling the handle), the main stream still waits for the side stream on the way out.
            main = torch.cuda.current_stream(self.engine.device)
            side.wait_stream(main)

            def joined():
                self._pending_join = None
                main.wait_stream(side)
            self._pending_join = joined                 # step() calls it on the way out if its body did not (early return, exception)
            try:
                with torch.cuda.stream(side):
                    yield joined
            except BaseException:
                joined()
                raise
        return cm()

    def _sum_over_ranks(self, flat):
        """DistributedDataParallel's gradient all-reduce for one flat arena (the reference wraps the generator and both discriminators,
        train_faceoff_perceptual.py:164-169 / disc trainer `modelD.module`): SUM in place behind the current stream, which then waits for it;
        returns the 1 / world the Adam launch folds in (DDP averages)."""
        if self.collectives:
            if self.comm is not None:
                self.comm.allreduce_async(flat)
                self.comm.wait()
            else:
                torch.distributed.all_reduce(flat)
            self.collectives_issued += 1
        return 1.0 / self.world

    def _broadcast_disc_buffers(self):
        """DDP's broadcast_buffers=True (the default the reference's wrap takes): module buffers that no collective sums -- the discriminators'
        InstanceNorm running statistics, each rank's moved by its OWN clip -- are overwritten with rank 0's at the start of every forward.
        Here: one broadcast of each discriminator's buffer arena at the END of the iteration (same state on rank 0, whose checkpoint is the
        one written; the other ranks hold rank 0's chain instead of diverging until their next forward).  Training-mode InstanceNorm never reads
        these statistics, so gradients do not depend on it."""
        if not self.collectives:
            return
        for d in (self.d3, self.d2):
            if self.comm is not None:
                self.comm.broadcast_async(d.flat_buffers, 0)
            else:
                torch.distributed.broadcast(d.flat_buffers, 0)
            self.collectives_issued += 1
        if self.comm is not None:
            self.comm.wait()

    # ------------------------------------------------------------------ the random choices, in the reference's call order
    def draw(self, num_frames, generator_iteration):
        r, w = self.rng, self.window
        c = {"random_idx": r.randint(0, num_frames - w)}                      # :330
        if generator_iteration:
            c["frame_id"] = r.randint(1, w - 1)                               # :351
            c["flip_real"] = r.randint(0, 1) == 0                             # flip_video(real) :368
            c["flip_fake"] = r.randint(0, 1) == 0                             # flip_video(fake) :369
        else:
            c["flip_fake"] = r.randint(0, 1) == 0                             # :392
            c["flip_real"] = r.randint(0, 1) == 0                             # :393
            c["frame_id"] = r.randint(1, w - 1)                               # :411
        return c

    # ------------------------------------------------------------------ discriminator inputs for one window
    def _video_pairs(self, dec_win, gt_win, c):
        """-> x [2][w-1][H][W][32]: sample 0 = fake pairs, sample 1 = real pairs."""
        w = self.window
        _, H, W, _ = dec_win.shape
        x = torch.empty((2, w - 1, H, W, 32), device=self.engine.device)
        for n, (src, nchw, flip) in enumerate(((dec_win, False, c["flip_fake"]), (gt_win, True, c["flip_real"]))):
            first, step = (w - 1, -1) if flip else (1, 1)
            make_pairs(src, nchw, 0, first, step, w - 1, x[n])
        return x

    def _image_pairs(self, dec_win, gt_win, c):
        _, H, W, _ = dec_win.shape
        x = torch.empty((2, 1, H, W, 32), device=self.engine.device)
        make_pairs(dec_win, False, 0, c["frame_id"], 1, 1, x[0])
        make_pairs(gt_win, True, 0, c["frame_id"], 1, 1, x[1])
        return x

    # ------------------------------------------------------------------ one iteration
    def step(self, img, ground_truth, choices=None, force_ids=None):
        """One GAN iteration (see _step).  Whatever happens inside -- an exception, an early return -- the image discriminator's side stream is
        joined into the main stream before control leaves: the tensors that cross between the two rely on that edge (see _beside)."""
        try:
            return self._step(img, ground_truth, choices, force_ids)
        finally:
            if getattr(self, "_pending_join", None) is not None:
                self._pending_join()

    def _step(self, img, ground_truth, choices=None, force_ids=None):
        """img [N,6,H,W] (one clip of N >= window frames, utils.py:29-38), ground_truth [N,3,H,W].  Returns a dict of
        device scalars; which keys depends on the iteration's parity (generator: recon, latent, g_loss_2d, g_loss_3d;
        discriminator: recon, latent, d_loss_3d, d_loss_2d)."""
        eng = self.engine
        gen_iter = self.iteration % 2 == 0                                    # :338-341
        self.iteration += 1
        N = img.shape[0]
        # the pairing kernel reads the ground truth through a raw pointer as dense [F,3,H,W] fp32: validate it ONCE here (a strided or
        # 4-channel tensor would pass the MSE ops, which densify their own copy, and then feed garbage 'real' pairs to both discriminators)
        ground_truth = ops.dense_f32(ground_truth, "ground truth")
        if ground_truth.dim() != 4 or ground_truth.shape[1] != 3 or ground_truth.shape[0] != N:
            raise ValueError(f"ground truth must be [N={N},3,H,W], got {tuple(ground_truth.shape)}")
        c = choices if choices is not None else self.draw(N, gen_iter)
        w, r = self.window, c["random_idx"]
        assert N >= w and 0 <= r <= N - w
        S = eng.forward(img, training=True, T=N, force_ids=force_ids)          # (force_ids: teacher-forced codes, parity tests: VQVAEEngine.forward)
        self.last_ids = (S["id_t"], S["id_b"])
        if getattr(self, "keep_states", False):                               # parity tests only: the generator's forward state stays alive
            self.last_gen_state = S                                           # (its ReLU branches are read back: tests/_fullsize_oracle.py)
        dec = S["dec"]
        acc = torch.empty(1, device=eng.device)                               # (overwritten by either MSE launch)
        if gen_iter:                                                          # loss value and gradient in one pass over dec and gt (as FaceOffTrainer.step)
            one = torch.ones(1, device=eng.device)
            g_dec = torch.empty_like(dec)
            ops.mse_slice_fwd_bwd(dec, ground_truth, acc, one, g_dec)
        else:
            ops.mse_slice_fwd(dec, ground_truth, acc)
        out = {"recon": acc / float(ground_truth.numel()), "latent": S["diff"]}
        dec_win, gt_win = dec[r:r + w], ground_truth[r:r + w]
        if gen_iter:
            g_win = g_dec[r:r + w]
            # image discriminator: module calls fake, then real (:354,357); only the fake logits reach the generator
            with self._beside() as joined:
                x2 = self._image_pairs(dec_win, gt_win, c)
                S2 = self.d2.forward(x2, training=True, sample_order=[0, 1])
                l2 = torch.zeros(1, device=eng.device)
                g2 = ralsgan_pair(S2["logits"], 0, 1, 1.0, 0.0, 0.5, l2, want_gb=False)
                gx2 = self.d2.backward(S2, g2, param_grads=False, input_grad=True, samples=(0, 1))    # (the real sample's logits carry no gradient)
            # video discriminator: module calls real, then fake (:368-369)
            x3 = self._video_pairs(dec_win, gt_win, c)
            S3 = self.d3.forward(x3, training=True, sample_order=[1, 0])
            l3 = torch.zeros(1, device=eng.device)
            g3 = ralsgan_pair(S3["logits"], 0, 1, 1.0, 0.0, 0.5, l3, want_gb=False)
            gx3 = self.d3.backward(S3, g3, param_grads=False, input_grad=True, samples=(0, 1))
            if getattr(self, "keep_states", False):
                self.last_disc_states = (S2, S3)                              # (what the discriminators kept for their backward: parity tests read the LeakyReLU masks)
            joined()                                                          # both input gradients add into g_win: on this stream, image then video
            pairs_backward(gx2[0], 0, c["frame_id"], 1, 1, g_win)
            first, step = (w - 1, -1) if c["flip_fake"] else (1, 1)
            pairs_backward(gx3[0], 0, first, step, w - 1, g_win)
            if getattr(self, "keep_states", False):
                self.last_g_dec = g_dec                                       # (d G_loss / d dec before the engine's backward consumes it: parity tests)
            eng.backward(S, g_dec, one * LATENT_LOSS_WEIGHT)                  # G_loss = recon + latent + G_2d + G_3d (:375)
            if self.scheduler is not None:
                self.scheduler.step()                                         # :380-381, before the optimiser
            self.optimizer.step(grad_scale=self._sum_over_ranks(eng.flat_grads))
            out.update(g_loss_2d=l2, g_loss_3d=l3)
        else:
            # image discriminator: module calls real, then fake (:412-413) -- its whole update beside the video discriminator's (the reference runs
            # the video discriminator first, :392-409; the two updates touch disjoint parameters and random draws were made above, in its order)
            with self._beside() as joined:
                x2 = self._image_pairs(dec_win, gt_win, c)
                S2 = self.d2.forward(x2, training=True, sample_order=[1, 0])
                l2 = torch.zeros(1, device=eng.device)
                g2 = ralsgan_pair(S2["logits"], 1, 0, 1.0, 0.0, 0.5, l2)
                self.d2.backward(S2, g2, param_grads=True, input_grad=False)
                self.d2.adam_step(self.d_lr, grad_scale=self._sum_over_ranks(self.d2.flat_grads))
            # video discriminator: module calls fake, then real (:392-393); both logits carry gradient to its parameters
            x3 = self._video_pairs(dec_win, gt_win, c)
            S3 = self.d3.forward(x3, training=True, sample_order=[0, 1])
            l3 = torch.zeros(1, device=eng.device)
            g3 = ralsgan_pair(S3["logits"], 1, 0, 1.0, 0.0, 0.5, l3)
            self.d3.backward(S3, g3, param_grads=True, input_grad=False)
            if getattr(self, "keep_states", False):
                self.last_disc_states = (S2, S3)
            self.d3.adam_step(self.d_lr, grad_scale=self._sum_over_ranks(self.d3.flat_grads))
            joined()
            out.update(d_loss_3d=l3, d_loss_2d=l2)
        self._broadcast_disc_buffers()
        return out
