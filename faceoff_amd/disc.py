"""MoCoGAN-HD multiscale PatchGAN discriminators on the gfx950 kernels (BASELINE config 5).

`DiscEngine(dims=3)` = ModelD_3d.netD, `DiscEngine(dims=2)` = ModelD_img.netD of the reference
(TemporalAlignment/models/mocoganhd_video_disc.py:56-176, mocoganhd_content_disc.py): num_D scales of
Conv k4 s2 p2 -> LeakyReLU | (Conv k4 s2 p2 -> InstanceNorm -> LeakyReLU) x2 | Conv k4 s1 p2 -> IN -> LReLU | Conv k4 s1 p2 -> 1,
the input of scale i+1 being the AvgPool(3, count_include_pad=False) of scale i's.

Like VQVAEEngine: an explicit forward that keeps what the backward needs (layer inputs, post-activation outputs, instance
statistics), an explicit backward, parameters and gradients in two flat arenas in reference state_dict order / shapes
(OIDHW), one Adam launch over the arena.  Real and fake inputs go through as a batch of samples (N = 2): InstanceNorm is
per sample, convolutions and filter gradients run once over both.

Layout: channels-last [N][D][H][W][C], input channels padded to 32 with zeros (the general conv kernel contracts 32-channel
chunks); the 1-channel patch logits live in a 32-float pixel (channel 0), which is also what their data gradient reads.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib, ops
import os as _os

from ._lib import ConvNdDesc, FO_BIAS, FO_ADD, FO_OUT_LRELU, FO_MASK_LRELU, FO_KSPLIT
from .synth import disc_param_specs, DISC_CHANNELS

STRIDES = (2, 2, 2, 1, 1)
PAD, K, SLOPE, IN_EPS, IN_MOMENTUM = 2, 4, 0.2, 1e-5, 0.1


def _out(n, s):
    return (n + 2 * PAD - K) // s + 1


class DiscEngine:
    def __init__(self, state_dict, device, dims=3, nc=6, num_D=2, n_frames=15):
        assert dims in (2, 3)
        self.device, self.dims, self.nc, self.num_D, self.n_frames = torch.device(device), dims, nc, num_D, n_frames
        self.specs = disc_param_specs(dims, nc, num_D)
        self._order_cache = {}
        sizes = [(k, s, int(torch.tensor(s).prod()) if len(s) else 1) for k, s in self.specs if k.endswith((".weight", ".bias"))]
        total = sum((n + 3) // 4 * 4 for _, _, n in sizes)
        self.flat_params = torch.zeros(total, device=self.device)
        self.flat_grads = torch.zeros(total, device=self.device)
        self.params, self.grads, off = OrderedDict(), OrderedDict(), 0
        for k, s, n in sizes:
            self.params[k] = self.flat_params[off:off + n].view(s)
            self.grads[k] = self.flat_grads[off:off + n].view(s)
            off += (n + 3) // 4 * 4
        self.buffers = OrderedDict()
        # every layer's [running_mean | running_var] in ONE arena: data parallel, rank 0's copy is broadcast in one message (GANTrainer)
        self.flat_buffers = torch.zeros(sum(2 * s[0] for k, s in self.specs if k.endswith("running_mean")), device=self.device)
        boff = 0
        for k, s in self.specs:
            if k.endswith("running_mean"):
                # [running_mean | running_var] of a layer live side by side (one kernel argument)
                base = k[:-len("running_mean")]
                both = self.flat_buffers[boff:boff + 2 * s[0]]
                boff += 2 * s[0]
                both[s[0]:] = 1.0
                self.buffers[base + "running_mean"] = both[:s[0]]
                self.buffers[base + "running_var"] = both[s[0]:]
                self.buffers[base + "_both"] = both
            elif k.endswith("num_batches_tracked"):
                self.buffers[k] = torch.zeros((), dtype=torch.int64, device=self.device)
        self.m = torch.zeros_like(self.flat_params)
        self.v = torch.zeros_like(self.flat_params)
        self.t = 0
        self._wp, self._wpt, self._w2 = {}, {}, {}
        self._packs_stale, self._packed_version = True, -1
        self.overlap_scales = not _os.environ.get("FACEOFF_NO_DISC_SCALE_OVERLAP")
        self._scale_stream = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ state
    def load_state_dict(self, sd):
        self._packs_stale = True
        for k, v in sd.items():
            k = k[len("module."):] if k.startswith("module.") else k
            t = torch.as_tensor(v).to(self.device)
            if k in self.params:
                self.params[k].copy_(t.float())
            elif k in self.buffers:
                self.buffers[k].copy_(t)
            else:
                raise KeyError(k)

    def state_dict(self):
        out = OrderedDict()
        for k, _ in self.specs:
            out[k] = self.params[k] if k in self.params else self.buffers[k]
        return out

    def adam_step(self, lr, betas=(0.5, 0.999), eps=1e-8, grad_scale=1.0):
        """torch.optim.Adam(netD.parameters(), lr, betas=(0.5, 0.999)) (mocoganhd_video_disc.py:24-26) as one launch."""
        self.t += 1
        ops.adam_flat(self.flat_params, self.flat_grads, self.m, self.v, lr, self.t, betas, eps, grad_scale)
        self.mark_params_dirty()              # (the launch writes the arena through a raw pointer: torch's version counter does not see it)

    # ------------------------------------------------------------------ helpers
    def _desc(self, N, src_dims, cs, ld_s, dst_dims, cd, ld_d, stride, flags=0, ld_mask=0):
        d = ConvNdDesc()
        d.N = N
        d.Ds, d.Hs, d.Ws = src_dims
        d.Dd, d.Hd, d.Wd = dst_dims
        d.Cs, d.ldS, d.Cd, d.ldD = cs, ld_s, cd, ld_d
        three = self.dims == 3
        d.KD, d.KH, d.KW = (K if three else 1), K, K
        d.sD, d.sH, d.sW = (stride if three else 1), stride, stride
        d.pD, d.pH, d.pW = (PAD if three else 0), PAD, PAD
        d.ldMask, d.flags, d.slope = ld_mask, flags, SLOPE
        return d

    def _desc0(self, N, src_dims, cs, ld_s, dst_dims, cd, ld_d, flags=0, ld_mask=0):
        """The first layer on the space-to-depth image: kernel 2, stride 1, padding 1."""
        d = self._desc(N, src_dims, cs, ld_s, dst_dims, cd, ld_d, 1, flags, ld_mask)
        three = self.dims == 3
        d.KD, d.KH, d.KW = (2 if three else 1), 2, 2
        d.pD, d.pH, d.pW = (1 if three else 0), 1, 1
        return d

    def _s2d(self, x, inverse_into=None):
        """x [N][D][H][W][32] (nc real channels) -> xs [N][D'][H'][W'][64 | 32]; inverse_into: scatter a gradient xs back into it."""
        N, D, H, W, ld = (inverse_into if inverse_into is not None else x).shape
        three = self.dims == 3
        Dp = (D + 1) // 2 if three else D
        ldxs = 64 if three else 32
        if inverse_into is None:
            xs = torch.empty((N, Dp, (H + 1) // 2, (W + 1) // 2, ldxs), device=self.device)
            _lib.call("fo_space_to_depth2", ops._ptr(x), ld, ops._ptr(xs), ldxs, N, D, H, W, self.nc, int(three), 0, ops._stream())
            return xs
        _lib.call("fo_space_to_depth2", ops._ptr(inverse_into), ld, ops._ptr(x), ldxs, N, D, H, W, self.nc, int(three), 1, ops._stream())
        return inverse_into

    def mark_params_dirty(self):
        """Tell the engine its parameter arena was written by something torch's version counter cannot see: a raw-pointer launch (fo_adam_flat,
        a communicator broadcast / all-reduce of parameters) or a `.data` alias.  Every such writer must call this; the next forward re-packs."""
        self._packs_stale = True

    def pack_filters(self):
        """Checkpoint-layout filters -> forward and data-gradient packs.  Re-packed only when the weights have changed since the last pack:
        the generator iteration (:338-341 alternates G / D by step parity) runs both discriminators on weights the previous iteration
        already packed -- 40 small launches, 0.4 ms, in front of the first discriminator convolution.  "Changed" = this engine's own Adam
        launch or load_state_dict (explicit flag), or any in-place torch op on the parameter arena or a view of it, e.g. the module
        mirror's torch.optim step (the arena's autograd version counter)."""
        ver = self.flat_params._version
        if not self._packs_stale and self._packed_version == ver and self._wp:
            return
        self._packs_stale, self._packed_version = False, ver
        taps = K ** self.dims
        for k, w in self.params.items():
            if not k.endswith(".weight"):
                continue
            O, I = w.shape[:2]
            if k.endswith("_layer0.0.weight"):
                # first layer (nc = 6 input channels): k4 s2 p2 over x == k2 s1 p1 over the space-to-depth image (8 | 4 phases
                # x 6 channels = 48 | 24 -> one or two 32-channel K chunks instead of 64 | 16 taps of a 6-in-32 padded pixel)
                ph, taps2 = (8, 8) if self.dims == 3 else (4, 4)
                w2 = self._w2.setdefault(k, torch.empty((O, ph * I, taps2), device=self.device))
                _lib.call("fo_s2d_filter", ops._ptr(w), ops._ptr(w2), O, I, K if self.dims == 3 else 1, 0, ops._stream())
                for store, tr in ((self._wp, 0), (self._wpt, 1)):
                    rows, cols = (ph * I, O) if tr else (O, ph * I)
                    n = ((rows + 63) // 64 * 64) * taps2 * ((cols + 31) // 32 * 32)
                    if k not in store or store[k].numel() != n:
                        store[k] = torch.empty(n, device=self.device)
                    _lib.call("fo_pack_convnd", ops._ptr(w2), ops._ptr(store[k]), O, ph * I, taps2, tr, ops._stream())
                continue
            for store, tr in ((self._wp, 0), (self._wpt, 1)):
                rows, cols = (I, O) if tr else (O, I)
                n = ((rows + 63) // 64 * 64) * taps * ((cols + 31) // 32 * 32)
                if k not in store or store[k].numel() != n:
                    store[k] = torch.empty(n, device=self.device)
                _lib.call("fo_pack_convnd", ops._ptr(w), ops._ptr(store[k]), O, I, taps, tr, ops._stream())

    def _dims_of(self, x):
        return tuple(x.shape[1:4])

    # ------------------------------------------------------------------ forward
    def forward(self, x, training=True, sample_order=None):
        """x [N][D][H][W][32] (channels >= nc zero; D = 1 for the image discriminator).  Returns S with S['logits'][i] =
        patch logits [N][Do][Ho][Wo][32] (channel 0) of result[i] in the reference's order (scale{num_D-1-i} on the input
        downsampled i times).  sample_order: the order in which the reference called the module on the samples (it moves
        the InstanceNorm running statistics once per call)."""
        assert x.dim() == 5 and x.shape[-1] == 32 and x.is_contiguous()
        self.pack_filters()
        N = x.shape[0]
        order = list(sample_order) if sample_order is not None else list(range(N))
        S = {"scales": [], "logits": [], "N": N, "training": training}
        # the scales are independent once their inputs exist (scale i+1 sees the average pool of scale i's input; separate filters, separate
        # running statistics): the pooled inputs first, then the coarser scales -- launches of a few dozen tiles -- on a side stream beside the
        # full-resolution one
        hs = [x]
        for i in range(1, self.num_D):
            hs.append(self._downsample(hs[-1]))
        side, main = self._side(), None
        if side is not None and self.num_D > 1:
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
        scales = [None] * self.num_D
        for i in list(range(1, self.num_D)) + [0]:
            ctx = torch.cuda.stream(side) if (main is not None and i > 0) else contextlib.nullcontext()
            with ctx:
                scales[i] = self._scale_fwd(hs[i], f"netD.scale{self.num_D - 1 - i}", training, order)
        if main is not None:
            main.wait_stream(side)
        S["scales"] = scales
        S["logits"] = [sc["feat"][4] for sc in scales]
        return S

    def _side(self):
        """The stream the coarser scales run on (None: no overlap -- CPU-less builds never get here; FACEOFF_NO_DISC_SCALE_OVERLAP=1; bench.py's
        per-kernel region, which wants every launch alone on the GPU)."""
        if not self.overlap_scales or self.device.type != "cuda":
            return None
        if self._scale_stream is None:
            self._scale_stream = torch.cuda.Stream(device=self.device)
        return self._scale_stream

    def _pool_args(self):
        if self.dims == 3:
            return (3, 2 if self.n_frames > 16 else 1, 2, 2)
        return (1, 1, 2, 2)

    def _downsample(self, h):
        N, D, H, W, Cc = h.shape
        kD, sD, sH, sW = self._pool_args()
        Do = (D - 1) // sD + 1 if kD == 1 else (D - 1) // sD + 1
        Ho, Wo = (H - 1) // sH + 1, (W - 1) // sW + 1
        out = torch.empty((N, Do, Ho, Wo, Cc), device=self.device)
        for n in range(N):
            _lib.call("fo_avgpool3_fwd", ops._ptr(h[n]), ops._ptr(out[n]), D, H, W, Cc, Cc, kD, sD, sH, sW, ops._stream())
        return out

    def _instnorm_ws(self, N, rows, co):
        """(workspace, bytes) of the chunked InstanceNorm launches (fo_instnorm_ws_bytes; FACEOFF_INSTNORM_ONE_LAUNCH=1: the one-workgroup-per-8-channels
        kernels): the current stream's scratch buffer (ops._workspace)."""
        nb = 0 if _os.environ.get("FACEOFF_INSTNORM_ONE_LAUNCH") else int(_lib.load().fo_instnorm_ws_bytes(N, C.c_int64(rows), co))
        if nb <= 0:
            return None, 0
        return ops._workspace(nb, self.device), nb

    def _scale_fwd(self, x, prefix, training, order):
        N = x.shape[0]
        # (cached: a host list -> device tensor copy is a synchronous H2D transfer that drains the stream -- 3 ms per call here)
        order_t = self._order_cache.get(tuple(order))
        if order_t is None:
            order_t = self._order_cache[tuple(order)] = torch.tensor(order, dtype=torch.int32, device=self.device)
        feat, stats, inp = [], [None] * 5, []
        h, cin = x, 32
        for j, (co, s) in enumerate(zip(DISC_CHANNELS, STRIDES)):
            key = f"{prefix}_layer{j}"
            sd = self._dims_of(h)
            dd = tuple((_out(n, s) if (self.dims == 3 or a > 0) else 1) for a, n in enumerate(sd))
            if j == 0:
                xs = self._s2d(x)
                y = torch.empty((N,) + dd + (co,), device=self.device)
                d = self._desc0(N, self._dims_of(xs), xs.shape[-1], xs.shape[-1], dd, co, co, FO_BIAS | FO_OUT_LRELU)
                self._convnd(d, 0, xs, self._wp[key + ".0.weight"], self.params[key + ".0.bias"], None, y)
                feat.append(y)
                inp += [xs, y]
                h, cin = y, co
                continue
            ld_out = max(32, co)
            # the 1-channel head is written into a zeroed 32-float pixel
            # few output tiles behind a long contraction -- the 1-channel head (192 tiles x 1024 K-steps) and the 256 -> 512 layer (292 tiles
            # of a 16 384-deep contraction on 512 workgroup slots: one round, 57 % full) --: the launch may slice K (through a workspace)
            # (and every other layer whose output is fewer than 1024 tiles: the epilogue is bias only, InstanceNorm is its own launch)
            few_tiles = (N * dd[0] * dd[1] * dd[2] + 63) // 64 * ((co + 63) // 64) < 1024
            ksplit = (co < 32 or j >= 3 or few_tiles) and not _os.environ.get("FACEOFF_NO_DISC_KSPLIT")
            y = (torch.zeros if co < 32 else torch.empty)((N,) + dd + (ld_out,), device=self.device)
            flags = FO_BIAS | (FO_OUT_LRELU if j == 0 else 0) | (FO_KSPLIT if ksplit else 0)
            d = self._desc(N, sd, cin, h.shape[-1], dd, co, ld_out, s, flags)
            if self._head_ok(co, s, cin):       # the 1-channel head: dot products, not a 64-column tile (csrc/disc_head.hip)
                self._profiled("fo_disc_head_fwd", self._conv_flops(d) if ops.PROFILER is not None else None, lambda: _lib.call(
                    "fo_disc_head_fwd", C.byref(d), ops._ptr(h), ops._ptr(self._wp[key + ".0.weight"]), ops._ptr(self.params[key + ".0.bias"]),
                    ops._ptr(y), ops._stream()))
            else:
                self._convnd(d, 0, h, self._wp[key + ".0.weight"], self.params[key + ".0.bias"], None, y)
            if 1 <= j <= 3:
                rows = dd[0] * dd[1] * dd[2]
                st = torch.empty((N, 2 * co), device=self.device)
                z = torch.empty_like(y)
                run = self.buffers[key + ".1._both"]
                ws, wsb = self._instnorm_ws(N, rows, co)
                _lib.call("fo_instnorm_lrelu_fwd_batch", ops._ptr(y), ld_out, ops._ptr(z), ld_out, N, C.c_int64(rows), co, C.c_float(IN_EPS),
                          C.c_float(SLOPE), ops._ptr(st), ops._ptr(run), ops._ptr(order_t), C.c_float(IN_MOMENTUM), int(not training), ops._ptr(ws),
                          C.c_int64(wsb), ops._stream())
                stats[j] = st
                y = z
            feat.append(y)
            inp.append(y)
            h, cin = y, max(32, co)
        return {"feat": feat, "stats": stats, "inp": inp[:5], "prefix": prefix, "x_shape": tuple(x.shape)}

    # ------------------------------------------------------------------ backward
    def backward(self, S, g_logits, param_grads=True, input_grad=False, samples=None):
        """g_logits[i]: gradient of the loss with respect to S['logits'][i] (same shape; only channel 0 is read).
        Fills self.grads (param_grads) and / or returns the gradient with respect to the input x [N][D][H][W][32].
        samples = (n0, n1): only samples n0 .. n1-1 carry gradient (the generator iteration: the real sample's logits get none, :359-360) --
        the backward runs on those samples alone and returns their input gradient [n1-n0][D][H][W][32]; not with param_grads."""
        n0, n1 = samples if samples is not None else (0, S["N"])
        assert 0 <= n0 < n1 <= S["N"] and (samples is None or not param_grads)
        N = n1 - n0
        gx_scale = [None] * self.num_D
        side, main = self._side(), None
        if side is not None and self.num_D > 1:
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
        for i in list(range(1, self.num_D)) + [0]:               # (as the forward: the coarser scales beside the full-resolution one)
            ctx = torch.cuda.stream(side) if (main is not None and i > 0) else contextlib.nullcontext()
            with ctx:
                gx_scale[i] = self._scale_bwd(S["scales"][i], g_logits[i], param_grads, input_grad, n0, n1)
        if main is not None:
            main.wait_stream(side)
        if not input_grad:
            return None
        # chain the scales: the input of scale i+1 is the average pool of the input of scale i
        kD, sD, sH, sW = self._pool_args()
        for i in range(self.num_D - 2, -1, -1):
            gx, gnext = gx_scale[i], gx_scale[i + 1]
            _, D, H, W, Cc = gx.shape
            for n in range(N):
                _lib.call("fo_avgpool3_bwd", ops._ptr(gnext[n]), ops._ptr(gx[n]), D, H, W, Cc, Cc, kD, sD, sH, sW, ops._stream())
        return gx_scale[0]

    @staticmethod
    def _head_ok(co, stride, cin):
        return co == 1 and stride == 1 and cin in (256, 512) and not _os.environ.get("FACEOFF_NO_DISC_HEAD")

    @staticmethod
    def _conv_flops(d, transposed=0, cs_real=None):
        """(algorithmic, nominal) FLOP of the convolution d describes, in either direction: 2 x output positions x Cout x taps x Cin, where
        `algorithmic` counts only the taps whose input exists (per axis, the (position, tap) pairs inside the input: padded taps are
        structural zeros the kernels skip) and `nominal` counts every tap.  Channels as the launch sees them (the source padded to 32)."""
        # the conv's input grid / output grid: forward = (source, destination), transposed = (destination, source)
        gin = (d.Dd, d.Hd, d.Wd) if transposed else (d.Ds, d.Hs, d.Ws)
        gout = (d.Ds, d.Hs, d.Ws) if transposed else (d.Dd, d.Hd, d.Wd)
        valid, nominal = 1.0, 1.0
        for nin, nout, k, st, pd in zip(gin, gout, (d.KD, d.KH, d.KW), (d.sD, d.sH, d.sW), (d.pD, d.pH, d.pW)):
            valid *= sum(1 for o in range(nout) for t in range(k) if 0 <= o * st + t - pd < nin)
            nominal *= nout * k
        ch = 2.0 * d.N * d.Cd * (cs_real if cs_real is not None else d.Cs)
        return ch * valid, ch * nominal

    @staticmethod
    def _profiled(label, flops, fn):
        prof = ops.PROFILER
        if prof is not None:
            prof.begin(label, *flops)
        fn()
        if prof is not None:
            prof.end()

    @classmethod
    def _convnd(cls, d, transposed, src, wp, bias, mask, dst):
        """fo_convnd with the workspace its K-slices need (FO_KSPLIT in d.flags and a shape the C side decides to slice)"""
        nbytes = _lib.load().fo_convnd_ws_bytes(C.byref(d), transposed) if d.flags & FO_KSPLIT else 0
        ws = ops._workspace(nbytes, src.device) if nbytes else None
        cls._profiled("fo_convnd" + ("_t" if transposed else ""), cls._conv_flops(d, transposed) if ops.PROFILER is not None else None, lambda: _lib.call(
            "fo_convnd", C.byref(d), transposed, ops._ptr(src), ops._ptr(wp), ops._ptr(bias), ops._ptr(mask), ops._ptr(dst),
            ops._ptr(ws), C.c_int64(ws.numel() * 4 if ws is not None else 0), ops._stream()))

    @classmethod
    def _wgradnd(cls, d, g, src, dw, cs_real):
        ws = ops._workspace(_lib.load().fo_wgradnd_ws_bytes(C.byref(d)), g.device)
        cls._profiled("fo_wgradnd", cls._conv_flops(d, 0, cs_real) if ops.PROFILER is not None else None, lambda: _lib.call(
            "fo_wgradnd", C.byref(d), ops._ptr(g), ops._ptr(src), ops._ptr(dw), cs_real, ops._ptr(ws), C.c_int64(ws.numel() * 4), ops._stream()))

    def _wgrad(self, d, g, src, key, cs_real):
        dw = self.grads[key + ".0.weight"]
        if self._head_ok(d.Cd, d.sH, d.Cs):
            ws = ops._workspace(_lib.load().fo_disc_head_wgrad_ws_bytes(C.byref(d)), g.device)
            self._profiled("fo_disc_head_wgrad", self._conv_flops(d, 0, cs_real) if ops.PROFILER is not None else None, lambda: _lib.call(
                "fo_disc_head_wgrad", C.byref(d), ops._ptr(g), ops._ptr(src), ops._ptr(dw), cs_real, ops._ptr(ws), C.c_int64(ws.numel() * 4), ops._stream()))
            rows = g.numel() // g.shape[-1]
            ops.bias_grad(g.view(rows, 1, 1, g.shape[-1]), self.grads[key + ".0.bias"], d.Cd)
            return
        self._wgradnd(d, g, src, dw, cs_real)
        rows = g.numel() // g.shape[-1]
        ops.bias_grad(g.view(rows, 1, 1, g.shape[-1]), self.grads[key + ".0.bias"], d.Cd)

    def _scale_bwd(self, sc, g_logit, param_grads, input_grad, n0, n1):
        prefix = sc["prefix"]
        cut = lambda ts: [None if t is None else t[n0:n1] for t in ts]       # (sample slices of channels-last tensors are contiguous)
        feat, stats, inp = cut(sc["feat"]), cut(sc["stats"]), cut(sc["inp"])
        N = n1 - n0
        g = g_logit[n0:n1]               # gradient wrt the conv output of layer j (pre-norm), channels-last, ld = max(32, co)
        for j in range(4, -1, -1):
            key = f"{prefix}_layer{j}"
            co, s = DISC_CHANNELS[j], STRIDES[j]
            x_in = inp[j]
            cin_pad = x_in.shape[-1]
            cin_real = self.nc if j == 0 else DISC_CHANNELS[j - 1]
            sd, dd = self._dims_of(x_in), self._dims_of(feat[j])
            if 1 <= j <= 3:              # g is wrt the post-activation output: through LeakyReLU and InstanceNorm
                rows = dd[0] * dd[1] * dd[2]
                gc = torch.empty_like(g)
                ws, wsb = self._instnorm_ws(N, rows, co)
                _lib.call("fo_instnorm_lrelu_bwd_batch", ops._ptr(g), g.shape[-1], ops._ptr(feat[j]), feat[j].shape[-1], ops._ptr(stats[j]),
                          ops._ptr(gc), gc.shape[-1], N, C.c_int64(rows), co, C.c_float(SLOPE), ops._ptr(ws), C.c_int64(wsb), ops._stream())
                g = gc
            if j == 0:
                # first layer on the space-to-depth image (x_in = xs): filter gradient in the k2 layout, mapped back to k4
                ph = 8 if self.dims == 3 else 4
                if param_grads:
                    d = self._desc0(N, sd, cin_pad, cin_pad, dd, co, g.shape[-1])
                    dw2 = torch.empty_like(self._w2[key + ".0.weight"])
                    self._wgradnd(d, g, x_in, dw2, ph * self.nc)
                    _lib.call("fo_s2d_filter", ops._ptr(self.grads[key + ".0.weight"]), ops._ptr(dw2), co, self.nc,
                              K if self.dims == 3 else 1, 1, ops._stream())
                    rows = g.numel() // g.shape[-1]
                    ops.bias_grad(g.view(rows, 1, 1, g.shape[-1]), self.grads[key + ".0.bias"], co)
                if not input_grad:
                    return None
                gxs = torch.empty_like(x_in)
                d = self._desc0(N, dd, co, g.shape[-1], sd, ph * self.nc, cin_pad)
                self._convnd(d, 1, g, self._wpt[key + ".0.weight"], None, None, gxs)
                gx = torch.zeros((N,) + tuple(sc["x_shape"][1:]), device=self.device)
                return self._s2d(gxs, inverse_into=gx)
            if param_grads:
                d = self._desc(N, sd, cin_pad, cin_pad, dd, co, g.shape[-1], s)
                self._wgrad(d, g, x_in, key, cin_real)
            # data gradient: source = g on the conv's output grid (channels padded to 32), destination = the conv's input
            cs = max(32, co)
            ksplit = j >= 3 and not _os.environ.get("FACEOFF_NO_DISC_KSPLIT")     # (as the forward: few tiles, long contraction, linear epilogue)
            gin = torch.empty_like(x_in) if j > 0 else torch.zeros_like(x_in)
            flags = (FO_MASK_LRELU if j == 1 else 0) | (FO_KSPLIT if ksplit else 0)          # layer 0's LeakyReLU (no norm in between): mask = its output
            if self._head_ok(co, s, cin_pad) and cin_real == cin_pad:
                dfw = self._desc(N, sd, cin_pad, cin_pad, dd, co, g.shape[-1], s)       # the forward convolution's description
                self._profiled("fo_disc_head_dgrad", self._conv_flops(dfw) if ops.PROFILER is not None else None, lambda: _lib.call(
                    "fo_disc_head_dgrad", C.byref(dfw), ops._ptr(g), ops._ptr(self._wp[key + ".0.weight"]), ops._ptr(gin), ops._stream()))
            else:
                d = self._desc(N, dd, cs, g.shape[-1], sd, cin_real, cin_pad, s, flags, ld_mask=cin_pad if j == 1 else 0)
                self._convnd(d, 1, g, self._wpt[key + ".0.weight"], None, x_in if j == 1 else None, gin)
            g = gin
        return g


# ---------------------------------------------------------------------------------------------------- loss + inputs
def ralsgan_pair(logits, ia, ib, target_a, target_b, weight, loss_acc, want_ga=True, want_gb=True, gscale=None):
    """For every scale's logits tensor [N][..][32] with samples a = logits[ia], b = logits[ib]:
        loss_acc += weight * (MSE(a - mean(b), target_a) + MSE(b - mean(a), target_b))
    i.e. Relativistic_Average_LSGAN both ways, summed over scales (mocoganhd_losses.py:113-126; trainer :359-360,372-373,
    404-408,416-420 with weight 0.5).  Returns per-scale gradients shaped like the logits (zero for samples not wanted)."""
    # fo_ralsgan adds into loss_acc with a plain read-modify-write from its only workgroup (no float atomic: the loss is the same bits every
    # run).  That is correct only while every launch sharing one accumulator is ordered on ONE stream -- asserted here, since the per-scale and
    # per-discriminator side streams of round 4 made the contract easy to break: an accumulator is bound to the stream it is first used on.
    st = torch.cuda.current_stream(loss_acc.device)
    bound = getattr(loss_acc, "_fo_stream", None)
    assert bound is None or bound == st, "ralsgan_pair: one loss accumulator used from two streams (its adds are not atomic)"
    loss_acc._fo_stream = st
    grads = []
    for lg in logits:
        g = torch.zeros_like(lg)
        n, ld = lg[0].numel() // lg.shape[-1], lg.shape[-1]
        _lib.call("fo_ralsgan", ops._ptr(lg[ia]), n, ops._ptr(lg[ib]), n, ld, C.c_float(target_a), C.c_float(target_b), C.c_float(weight),
                  ops._ptr(loss_acc), ops._ptr(gscale), ops._ptr(g[ia]) if want_ga else None, ops._ptr(g[ib]) if want_gb else None,
                  ops._stream())
        grads.append(g)
    return grads


def make_pairs(frames, nchw, f0, first, step, n, out):
    """out[j] = (frame f0 | frame first + j*step) channels 0..5 of a 32-float pixel (see fo_disc_pairs)."""
    if nchw:
        F, _, H, W = frames.shape
        ld = 0
    else:
        F, H, W, ld = frames.shape
    _lib.call("fo_disc_pairs", ops._ptr(frames), int(nchw), ld, H, W, f0, first, step, n, ops._ptr(out), 32, ops._stream())
    return out


def pairs_backward(g_pairs, f0, first, step, n, g_frames, scale=1.0):
    """Adds the gradient of `make_pairs` into g_frames [F][H][W][ld] channels 0..2."""
    _, H, W, ld = g_frames.shape
    _lib.call("fo_disc_pairs_bwd", ops._ptr(g_pairs), 32, H, W, f0, first, step, n, ops._ptr(g_frames), ld, C.c_float(scale), ops._stream())
