"""The FaceOff VQ-VAE training step as a planned sequence of gfx950 kernel launches.

This is the MI355X-native counterpart of what torch autograd + cuDNN do for the reference's
`VQVAE.forward` (models/vqvae_conv3d_latent.py:243-285) and `loss.backward()`
(train_faceoff_perceptual.py:100): an explicit forward that keeps exactly the activations the
backward needs, and an explicit backward in reverse-forward order that emits every parameter
gradient into one flat arena (so data-parallel buckets are plain slices of it and become ready
mid-backward).  No tracing, no graph compiler: the launch list is static Python over the C ABI.

Layout: channels-last everywhere; frames (B*T) outermost, so the reference's
`[N,C,H,W] <-> [1,C,N,H,W]` permutes (:247,251) are views.  Every ReLU, bias, residual add,
ReLU-backward mask, gradient fan-in and torch.cat of the reference is an epilogue flag or a
channel-slice view, never a standalone pass.
"""
from __future__ import annotations

from collections import OrderedDict

import os as _os

import torch

from . import ops
from .ops import FO_IN_RELU, FO_OUT_RELU
from .synth import vqvae_param_specs


def _rb(prefix):
    return [prefix + ".conv.3", prefix + ".conv.1"]


# The order in which VQVAEEngine.backward() finishes each layer's filter/bias gradient.
BACKWARD_ORDER = (
    ["dec.blocks.6", "dec.blocks.4"] + _rb("dec.blocks.2") + _rb("dec.blocks.1") + ["dec.blocks.0", "upsample_t",
     "quantize_conv_b", "dec_t.blocks.4"] + _rb("dec_t.blocks.2") + _rb("dec_t.blocks.1") + ["dec_t.blocks.0",
     "quantize_conv_t"] + [f"conv3d_encoded_t.conv3d.{i}.0" for i in (2, 1, 0)] + _rb("enc_t.blocks.4")
    + _rb("enc_t.blocks.3") + ["enc_t.blocks.2", "enc_t.blocks.0"]
    + [f"conv3d_encoded_b.conv3d.{i}.0" for i in (2, 1, 0)] + _rb("enc_b.blocks.6") + _rb("enc_b.blocks.5")
    + ["enc_b.blocks.4", "enc_b.blocks.2", "enc_b.blocks.0"])


WINO2D_MIN_CIN = int(_os.environ.get("FACEOFF_WINO2D_MIN_CIN", "64"))    # 3x3 Conv2d with >= this many input channels: F(4x4,3x3) (64: enc_t.2, dec_t.0; -0.13 ms per step)


class _Layer:
    """One conv-like layer: checkpoint-layout parameters + packed filters + launch helpers."""

    def __init__(self, name, kind, w, b, gw, gb):
        self.name, self.kind = name, kind
        self.w, self.b, self.gw, self.gb = w, b, gw, gb
        if kind == "convT":
            self.ci, self.co = w.shape[0], w.shape[1]
        else:
            self.co, self.ci = w.shape[0], w.shape[1]
        self.k = tuple(w.shape[2:])
        self.cip, self.cop = ops.pad_in(self.ci), ops.pad_in(self.co) if self.co < 32 else self.co
        self.wp = None     # forward filter
        self.wpd = None    # dgrad filter
        self.need_dgrad = True
        self.engine = None
        self._w42_buf, self._wino_buf = {}, {}   # transformed filter banks: persistent buffers, refilled by pack() once a step has used them

    # -- filter packing (every step: the optimiser rewrites the checkpoint-layout weights)
    def pack(self):
        # Winograd filter banks of this step: the forms the last step used are refilled here (on the pack stream, beside the previous
        # launches), a form not seen before is made on first use
        self._w42_u = {t: ops.w42_filter(self.w, t, out=U) for t, U in self._w42_buf.items()}
        if self.kind == "convT":
            if self.co <= 8:                         # few output channels: all 4 phases as one 3x3 filter bank
                self.wp = ops.pack_convT_fused(self.w, self.wp)
            else:
                self.wp = ops.pack_convT(self.w, self.wp)
            if self.need_dgrad:                      # dgrad = conv k4s2p1 with O:=ci, I:=co
                self.wpd = ops.pack_conv(self.w, self.wpd)
        else:
            self.wp = ops.pack_conv(self.w, self.wp)
            self._wino_u = {key: ops.wino_filter(self.w, dgrad=key[1], m=key[0], out=U) for key, U in self._wino_buf.items()}   # (m, dgrad) -> U
            if not self.need_dgrad:
                return
            if self.k[-1] == 4:                      # dgrad of k4s2p1 conv = transposed conv, Ci_T:=co, Co_T:=ci
                self.wpd = ops.pack_convT(self.w, self.wpd)
            else:
                self.wpd = ops.pack_conv_dgrad(self.w.reshape(self.co, self.ci, -1), self.wpd)

    def _geom(self):
        if self.kind == "conv3d":
            return dict(k=(3, 3, 3), pad=(1, 1, 1), stride=1)
        kk = self.k[-1]
        if kk == 4:
            return dict(k=(1, 4, 4), pad=(0, 1, 1), stride=2)
        return dict(k=(1, kk, kk), pad=(0, kk // 2, kk // 2), stride=1)

    # -- forward
    def fwd(self, x, out, T=1, flags=0, add=None):
        if self.kind == "convT" and self.co <= 8:
            assert add is None
            ops.convT_fused(x, self.wp, self.b, out, cin=self.cip, cout=self.co, flags=flags)
        elif self.kind == "convT" and self._w42("convT", x.shape[0], x.shape[1], x.shape[2], self.ci, self.co):
            ops.convT_k4s2_winograd(x, self._w42_filter(True), self.b, out, cin=self.ci, cout=self.co, flags=flags, add=add)
        elif self.kind == "convT":
            ops.convT_phases(x, self.wp, self.b, out, cin=self.cip, cout=self.co, flags=flags, add=add)
        elif self.k[-1] == 4 and self._w42("conv", x.shape[0], x.shape[1], x.shape[2], self.ci, self.co):
            S = self.engine._cur_S
            keep = S is not None and self.engine.keep_wino_v and ops.w42_wgrad_ok(x.shape[0], x.shape[1], x.shape[2], self.ci, self.co)
            V = ops.conv_k4s2_winograd(x, self._w42_filter(False), self.b, out, cin=self.ci, cout=self.co, flags=flags, add=add, keep_v=keep)
            if keep:
                S.setdefault("_wino_v", {})[self.name] = V
        elif self._winograd_m(x):
            m = self._winograd_m(x)
            kd = self._wino_kd
            # the transformed input is kept for this layer's filter gradient IN THE FORWARD'S OWN STATE DICT (two
            # forwards followed by the backward of the older one must not mix them up)
            S = self.engine._cur_S
            keep = (S is not None and self.engine.keep_wino_v
                    and ops.wino_wgrad_ok(x.shape[1], x.shape[2], x.shape[0], T, m, kd))
            V = ops.conv3d_winograd(x, self._wino_filter(m, False), self.b, out, T=T, cin=self.ci, cout=self.co,
                                    flags=flags, add=add, keep_v=keep, m=m, kd=kd)
            if keep:
                S.setdefault("_wino_v", {})[self.name] = V
        else:
            g = self._geom()
            ops.conv_igemm(x, self.wp, self.b, out, T=T if self.kind == "conv3d" else 1, cin=self.cip, cout=self.co,
                           flags=flags, add=add, **g)

    def _winograd_m(self, x, dgrad=False):
        """Output-tile size of the Winograd form for a Conv3d k3 p1 on frames like x: 4 (64x64 latents: 4x fewer MFMA
        FLOP), 2 (other even sizes: 2.25x fewer) or 0 = direct kernel (odd sizes, other layer kinds, FACEOFF_NO_WINOGRAD)."""
        eng = self.engine
        if eng is None or not eng.winograd or self.ci % 32 or self.co % 4:
            return 0
        m = min(ops.wino_tile(x.shape[1], x.shape[2], x.shape[0]), eng.winograd_max_tile)
        if self.kind == "conv3d":
            return m
        # Conv2d 3x3 s1 128->128 (enc_b.blocks.4, dec.blocks.0): one depth tap, K = Cin only -- pays with F(4x4) alone
        if self.kind == "conv" and self.k == (3, 3) and self.ci >= WINO2D_MIN_CIN and self.co >= 128 and m == 4:
            # (the data gradient's GEMM has Cout := ci columns: the plane-stack GEMM kernel wants whole 128-column tiles)
            return 4 if (not dgrad or self.ci % 128 == 0) else 0
        return 0

    def _w42(self, form, N, H, W, cin, cout, which="fwd"):
        """Winograd F(4x4, 2x2) form of a k4 s2 p1 stem pass (2.56x fewer MFMA FLOP): `form` = "conv" (Conv2d forward, ConvTranspose2d
        data gradient: [N,H,W,cin] -> [N,H/2,W/2,cout]), "convT" ([N,H,W,cin] -> [N,2H,2W,cout]) or "wgrad" (conv-form geometry).
        Measured per layer (tools/bench_w42.py): pays on enc_b.2, dec.4 and dec_t.4 (128 channels on the pixel-grid side: -20 % forward / data
        gradient, -25...-48 % filter gradient), not on upsample_t (64 -> 64) or enc_t.0 (64 channels on that side)."""
        eng = self.engine
        wide, narrow = (self.ci, self.co) if self.kind == "convT" else (self.co, self.ci)     # the cell side has 4 x narrow channels
        if eng is None or not eng.winograd or not eng.w42 or wide < 128 or narrow < 64 or (self.name, which) in eng.w42_skip:
            return False
        return {"conv": ops.w42_conv_ok, "convT": ops.w42_convT_ok, "wgrad": ops.w42_wgrad_ok}[form](N, H, W, cin, cout)

    def _w42_filter(self, transposed):
        if transposed not in self._w42_u:
            self._w42_u[transposed] = self._w42_buf[transposed] = ops.w42_filter(self.w, transposed)
        return self._w42_u[transposed]

    @property
    def _wino_kd(self):
        return 3 if self.kind == "conv3d" else 1

    def _wino_filter(self, m, dgrad):
        key = (m, dgrad)
        if key not in self._wino_u:
            self._wino_u[key] = self._wino_buf[key] = ops.wino_filter(self.w, dgrad=dgrad, m=m)
        return self._wino_u[key]

    # -- data gradient: gin = dgrad(g) [* (mask > 0)] [+ add]
    def dgrad(self, g, gin, T=1, mask=None, add=None):
        if self.kind == "convT" and self._w42("conv", g.shape[0], g.shape[1], g.shape[2], self.co, self.ci, "dgrad"):
            ops.conv_k4s2_winograd(g, self._w42_filter(False), None, gin, cin=self.co, cout=self.ci, mask=mask, add=add)
        elif self.kind == "convT":                   # conv k4 s2 p1 over g
            ops.conv_igemm(g, self.wpd, None, gin, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=ops.pad_in(self.co),
                           cout=self.ci, mask=mask, add=add)
        elif self.k[-1] == 4 and self._w42("convT", g.shape[0], g.shape[1], g.shape[2], self.co, self.ci, "dgrad"):
            ops.convT_k4s2_winograd(g, self._w42_filter(True), None, gin, cin=self.co, cout=self.ci, mask=mask, add=add)
        elif self.k[-1] == 4:                        # transposed conv over g
            ops.convT_phases(g, self.wpd, None, gin, cin=self.co, cout=self.ci, mask=mask, add=add)
        elif self._winograd_m(g, dgrad=True):
            m = self._winograd_m(g, dgrad=True)
            ops.conv3d_winograd(g, self._wino_filter(m, True), None, gin, T=T, cin=self.co, cout=self.ci, mask=mask, add=add, m=m,
                                kd=self._wino_kd)
        else:
            geo = self._geom()
            pad = tuple(kk - 1 - p for kk, p in zip(geo["k"], geo["pad"]))
            ops.conv_igemm(g, self.wpd, None, gin, T=T if self.kind == "conv3d" else 1, k=geo["k"], stride=1, pad=pad,
                           cin=self.co, cout=self.ci, mask=mask, add=add)

    # -- filter + bias gradient
    def wgrad(self, x, g, T=1, in_relu=False):
        """Enqueued on the engine's side stream when wgrad overlap is on: a layer's filter gradient and its data
        gradient both depend only on g, and an igemm workgroup (73 KB LDS) and a wgrad workgroup (64 KB) fit one CU
        together, so each kernel's tail and prologue are filled by the other's workgroups."""
        eng = self.engine
        if eng is not None and eng.wgrad_stream is not None and eng.defer_wgrad:
            # deferred: launched behind the NEXT Winograd-domain GEMM of the calling stream (ops.AFTER_GEMM), so that it starts
            # when that GEMM ends -- beside the HBM-bound output / input transforms that follow -- instead of beside the GEMM
            eng._keepalive.append((x, g))
            eng._pending_wgrad.setdefault(torch.cuda.current_stream().cuda_stream, []).append((self, x, g, T, in_relu))
        elif eng is not None and eng.wgrad_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            eng.wgrad_stream.wait_event(ev)
            eng._keepalive.append((x, g))          # allocated on the main stream, read on the side stream
            with torch.cuda.stream(eng.wgrad_stream):
                self._wgrad(x, g, T, in_relu)
                eng._ready(self.name)
        else:
            self._wgrad(x, g, T, in_relu)
            if eng is not None:
                eng._ready(self.name)

    def _wgrad(self, x, g, T, in_relu):
        geo = self._geom()
        if self.kind == "convT" and self._w42("wgrad", g.shape[0], g.shape[1], g.shape[2], self.co, self.ci, "wgrad"):
            # the adjoint convolution's filter gradient: its input is this layer's output gradient, its output gradient this layer's input
            ops.conv_k4s2_wgrad_winograd(g, x, self.gw, cin=self.co, cout=self.ci)
            ops.bias_grad(g, self.gb, self.co)
        elif self.kind == "convT":
            ops.conv_wgrad(x, g, self.gw, None, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=self.ci, b_real=self.co,
                           in_relu=False)
            ops.bias_grad(g, self.gb, self.co)
        elif self.k[-1] == 4 and not in_relu and self._w42("wgrad", x.shape[0], x.shape[1], x.shape[2], self.ci, self.co, "wgrad"):
            S = self.engine._cur_S
            V = S.get("_wino_v", {}).pop(self.name, None) if S is not None else None
            if V is not None and self.engine.wgrad_stream is not None:
                self.engine._keepalive.append(V)
            ops.conv_k4s2_wgrad_winograd(x, g, self.gw, cin=self.ci, cout=self.co, V=V, dbias=self.gb)
        elif (self._winograd_m(x) and not in_relu and self.ci == self.cip
              and ops.wino_wgrad_ok(x.shape[1], x.shape[2], x.shape[0], T, self._winograd_m(x), self._wino_kd)):
            S = self.engine._cur_S if self.engine is not None else None
            V = S.get("_wino_v", {}).pop(self.name, None) if S is not None else None   # the forward's transformed input, if kept
            if V is not None and self.engine is not None and self.engine.wgrad_stream is not None:
                self.engine._keepalive.append(V)                      # read on the side stream: must outlive this call
            ops.conv3d_wgrad_winograd(g, x, self.gw, self.gb, T=T, a_real=self.co, b_real=self.ci, V=V, m=self._winograd_m(x),
                                      kd=self._wino_kd)
        else:
            ops.conv_wgrad(g, x, self.gw, self.gb, T=T if self.kind == "conv3d" else 1, a_real=self.co, b_real=self.ci,
                           in_relu=in_relu, **geo)


BF16_CONVT_CELLS = not _os.environ.get("FACEOFF_BF16_CONVT_PHASES")     # (read at import, like ops.CONVT_CELLS: the packs and the launches must agree)


class _LayerBF16(_Layer):
    """The same layer with bf16 MFMA operands (BASELINE config 3 as SURVEY.md section 8(d) defines it): the fp32 master filter is packed
    and rounded to bf16 every step, activations and their gradients are bf16 tensors, accumulation, bias, filter / bias gradients fp32.
    Direct convolutions only (the bf16 matrix pipe is 16x the fp32 one: transforms would cost more than they save, and F(4x4) is not
    safe at 8 significand bits)."""

    def pack(self):
        def r16(f32, key):
            buf = getattr(self, key, None)
            out = ops.to_bf16(f32, buf)
            setattr(self, key, out)
            return out
        cells = BF16_CONVT_CELLS                  # k4 s2 transposed forms as one cell-form launch instead of four sub-pixel phases (round 6)
        if self.kind == "convT":
            self._p32 = (ops.pack_convT_fused if self.co <= 8 else (ops.pack_convT_cells if cells else ops.pack_convT))(self.w, getattr(self, "_p32", None))
            self.wp = r16(self._p32, "_wp16")
            if self.need_dgrad:                      # dgrad = conv k4s2p1 with O:=ci, I:=co
                self._pd32 = ops.pack_conv(self.w, getattr(self, "_pd32", None))
                self.wpd = r16(self._pd32, "_wpd16")
        else:
            self._p32 = ops.pack_conv(self.w, getattr(self, "_p32", None))
            self.wp = r16(self._p32, "_wp16")
            if not self.need_dgrad:
                return
            if self.k[-1] == 4:                      # dgrad of k4s2p1 conv = transposed conv, Ci_T:=co, Co_T:=ci
                self._pd32 = (ops.pack_convT_cells if cells else ops.pack_convT)(self.w, getattr(self, "_pd32", None))
            else:
                self._pd32 = ops.pack_conv_dgrad(self.w.reshape(self.co, self.ci, -1), getattr(self, "_pd32", None))
            self.wpd = r16(self._pd32, "_wpd16")

    def fwd(self, x, out, T=1, flags=0, add=None):
        if self.kind == "convT" and self.co <= 8:
            assert add is None
            ops.convT_fused_bf16(x, self.wp, self.b, out, cin=self.cip, cout=self.co, flags=flags)
        elif self.kind == "convT":
            (ops.convT_cells_bf16 if BF16_CONVT_CELLS else ops.convT_phases_bf16)(x, self.wp, self.b, out, cin=self.cip, cout=self.co, flags=flags, add=add)
        else:
            ops.conv_bf16g(x, self.wp, self.b, out, T=T if self.kind == "conv3d" else 1, cin=self.cip, cout=self.co, flags=flags, add=add,
                           **self._geom())

    def dgrad(self, g, gin, T=1, mask=None, add=None):
        if self.kind == "convT":                     # conv k4 s2 p1 over g
            ops.conv_bf16g(g, self.wpd, None, gin, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=ops.pad_in(self.co), cout=self.ci, mask=mask, add=add)
        elif self.k[-1] == 4:                        # transposed conv over g
            (ops.convT_cells_bf16 if BF16_CONVT_CELLS else ops.convT_phases_bf16)(g, self.wpd, None, gin, cin=self.co, cout=self.ci, mask=mask, add=add)
        else:
            geo = self._geom()
            pad = tuple(kk - 1 - p for kk, p in zip(geo["k"], geo["pad"]))
            ops.conv_bf16g(g, self.wpd, None, gin, T=T if self.kind == "conv3d" else 1, k=geo["k"], stride=1, pad=pad, cin=self.co, cout=self.ci,
                           mask=mask, add=add)

    def _wgrad(self, x, g, T, in_relu):
        if self.kind == "convT":                     # the adjoint convolution's filter gradient: P = this layer's input, Q = its output gradient
            ops.conv_wgrad_bf16(x, g, self.gw, None, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=self.ci, b_real=self.co)
            ops.bias_grad_bf16(g, self.gb, self.co)
        else:
            ops.conv_wgrad_bf16(g, x, self.gw, self.gb, T=T if self.kind == "conv3d" else 1, a_real=self.co, b_real=self.ci, in_relu=in_relu,
                                **self._geom())


def _runs_beside(stream, dev, busy=None):
    """True when a launch on `stream` finishes while long launches occupy `busy` (default: the current stream) -- i.e. the two do not share a hardware queue
    (packets of one queue are dispatched in order: the second would start when the first has handed out its last workgroup).  ~1 ms."""
    busy = busy if busy is not None else torch.cuda.current_stream(dev)
    buf = torch.empty(1 << 27, device=dev)                      # 512 MB: one elementwise pass over it is ~0.08 ms of a many-workgroup launch
    beside = False
    for _ in range(2):                                          # (the first round warms the launch paths up; the second one counts)
        t0, t_side, t_busy = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(busy):
            t0.record(busy)
            for _ in range(4):
                buf.fill_(1.0)
            t_busy.record(busy)
        with torch.cuda.stream(stream):
            torch.zeros(1, device=dev)
            t_side.record(stream)
        torch.cuda.synchronize(dev)
        beside = t0.elapsed_time(t_side) < 0.5 * t0.elapsed_time(t_busy)     # done well before the long launches were (measured: 0.06 against 0.33 ms)
    return beside


def _runs_beside_current(stream, dev):
    return _runs_beside(stream, dev)


_QUEUE_CLASSES = {}          # device index -> the result of the first streams_by_queue() of the process: the same streams serve every trainer after it


def streams_by_queue(dev, classes=3, per_class=3, max_draw=24):
    """Pooled streams sorted by the hardware queue HIP gave them: `classes` lists of `per_class` streams each, every list on its own queue, none on the current
    stream's; plus the streams that were found on the current stream's queue.  Queues are told apart by _runs_beside (a process cannot ask for a stream's queue).
    None when the pool does not yield that many (fewer hardware queues than expected).  Sorted once per process and device (~50 ms, a transient 512 MB): later
    trainers get the same streams -- a stream is an ordering, two engines that use one are merely ordered against each other where they would have run together."""
    key = (torch.device(dev).index or 0, classes, per_class)
    if key in _QUEUE_CLASSES:
        return _QUEUE_CLASSES[key]
    _QUEUE_CLASSES[key] = _sort_streams(dev, classes, per_class, max_draw)
    return _QUEUE_CLASSES[key]


def _sort_streams(dev, classes, per_class, max_draw):
    main = torch.cuda.current_stream(dev)
    groups, on_main = [], []
    for _ in range(max_draw):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)                          # (first use: the stream takes its queue)
        if not _runs_beside(st, dev, main):
            on_main.append(st)
            continue
        for g in groups:
            if not _runs_beside(st, dev, g[0]):
                g.append(st)
                break
        else:
            groups.append([st])
        full = [g for g in groups if len(g) >= per_class]
        if len(full) >= classes:
            return [g[:per_class] for g in full[:classes]], on_main
    return None


class VQVAEEngine:
    """Owns the flat parameter / gradient arenas and runs forward / backward on `device`."""

    def __init__(self, state_dict, device, in_channel=6, clip_len=None, dtype=None):
        """dtype: "fp32" (BASELINE config 2: fp32 MFMA, the reference's arithmetic) or "bf16" (config 3: bf16 MFMA operands, fp32 accumulation,
        fp32 master weights, fp32 VQ; rounding points = oracle.faceoff_oracle._BF16Sim).  Default: $FACEOFF_DTYPE, else fp32."""
        self.device = torch.device(device)
        self.in_channel = in_channel
        self.clip_len = clip_len
        self.dtype = dtype or _os.environ.get("FACEOFF_DTYPE", "fp32")
        assert self.dtype in ("fp32", "bf16"), self.dtype
        self.bf16 = self.dtype == "bf16"
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float32
        specs = vqvae_param_specs(in_channel=in_channel)
        # ---- flat arenas.  Arena order = REVERSE of the order in which backward() completes layers, so
        # the gradient arena fills from its end towards its start and every data-parallel bucket is one
        # contiguous slice that becomes ready mid-backward.  (state_dict() re-emits reference order.)
        by_name = {name: (kind, shape) for name, kind, shape in specs if kind != "vq"}
        assert set(by_name) == set(BACKWARD_ORDER), "BACKWARD_ORDER must list every layer exactly once"
        self.layer_order = list(reversed(BACKWARD_ORDER))
        sizes = []
        for name in self.layer_order:
            kind, shape = by_name[name]
            n_w = int(torch.tensor(shape).prod())
            n_b = shape[1] if kind == "convT" else shape[0]
            sizes += [(name + ".weight", shape, n_w), (name + ".bias", (n_b,), n_b)]
        total = sum((n + 3) // 4 * 4 for _, _, n in sizes)
        self.flat_params = torch.zeros(total, device=self.device)
        self.flat_grads = torch.zeros(total, device=self.device)
        self.params, self.grads, self.offsets = OrderedDict(), OrderedDict(), OrderedDict()
        off = 0
        for key, shape, n in sizes:
            self.params[key] = self.flat_params[off:off + n].view(shape)
            self.grads[key] = self.flat_grads[off:off + n].view(shape)
            self.offsets[key] = (off, n)
            off += (n + 3) // 4 * 4
        self.buffers = OrderedDict()
        for lvl in ("t", "b"):
            self.buffers[f"quantize_{lvl}.embed"] = torch.zeros(64, 512, device=self.device)
            self.buffers[f"quantize_{lvl}.cluster_size"] = torch.zeros(512, device=self.device)
            self.buffers[f"quantize_{lvl}.embed_avg"] = torch.zeros(64, 512, device=self.device)
        self.layers = OrderedDict()
        for name, kind, shape in specs:
            if kind == "vq":
                continue
            self.layers[name] = (_LayerBF16 if self.bf16 else _Layer)(name, kind, self.params[name + ".weight"], self.params[name + ".bias"],
                                                                      self.grads[name + ".weight"], self.grads[name + ".bias"])
        self.layers["enc_b.blocks.0"].need_dgrad = False   # the input image needs no gradient
        for layer in self.layers.values():
            layer.engine = self
        self.wgrad_stream = None
        if self.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_WGRAD_OVERLAP"):
            self.wgrad_stream = torch.cuda.Stream(device=self.device)
        # second side stream: the bottom Conv3d chain (forward and backward) is independent of the top-level chain
        # (enc_t / quantize_t / dec_t), whose launches are small (1280 tiles); run side by side, each fills the
        # other's tails -- two igemm workgroups per CU fit whichever kernel they come from.
        self.aux_stream = None
        if self.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_CHAIN_OVERLAP"):
            self.aux_stream = torch.cuda.Stream(device=self.device)
        # third side stream: the ~60 filter-packing launches of a step (tiny, latency-bound) beside the input layout kernel and
        # the first two layers instead of in front of them (tools/timeline.py: they were 1.3 ms of nothing-but-small-kernels)
        self.pack_stream = None
        if self.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_PACK_OVERLAP"):
            self.pack_stream = torch.cuda.Stream(device=self.device)
        # fourth side stream: the quantisers' EMA statistics, their cross-rank sum and the codebook update.  Nothing in the step reads them
        # (the NEXT forward's vq_prepare does): off the forward's critical path, where they were 0.4 ms of HBM-bound work -- and, with more
        # than one rank, a blocking all-reduce
        self.vq_stream = None
        if self.device.type == "cuda" and not _os.environ.get("FACEOFF_NO_VQ_OVERLAP"):
            self.vq_stream = torch.cuda.Stream(device=self.device)
        self._vq_done = None        # event behind the last codebook update on vq_stream (_quantize)
        self._streams = (self.wgrad_stream, self.aux_stream, self.pack_stream, self.vq_stream)
        self.tail_wgrads = int(_os.environ.get("FACEOFF_TAIL_WGRADS", "2"))     # how many of the last filter gradients run on the caller's stream (backward())
        self._pack_events = None
        self._packs_stale, self._packed_version = True, -1      # pack_filters(): skip when the weights have not changed since the last pack
        # Conv3d forward / data gradient as Winograd F(2x2,3x3) + a (3,1,1) implicit GEMM (FACEOFF_NO_WINOGRAD=1: direct)
        self.winograd = not _os.environ.get("FACEOFF_NO_WINOGRAD")
        self.winograd_max_tile = int(_os.environ.get("FACEOFF_WINOGRAD_TILE", "4"))   # 2: F(2x2,3x3) everywhere
        self.w42 = not _os.environ.get("FACEOFF_NO_W42")       # k4 s2 stems as Winograd F(4x4, 2x2) (needs self.winograd too)
        # (layer, pass) pairs kept OFF the F(4x4,2x2) form, pass in {"fwd", "dgrad", "wgrad"}
        self.w42_skip = {tuple(it.split(":")) for it in _os.environ.get("FACEOFF_W42_SKIP", "").split(",") if it}
        self.fused_resblock = not _os.environ.get("FACEOFF_NO_FUSED_RESBLOCK")
        self.fused_resblock_bwd = self.fused_resblock and not _os.environ.get("FACEOFF_NO_FUSED_RESBLOCK_BWD")
        self.keep_wino_v = False      # training forward: keep each Conv3d's transformed input for its filter gradient
        # filter gradients start when the next Winograd-domain GEMM of their stream ENDS (beside the HBM-bound transforms that
        # follow it) rather than beside that GEMM: -0.33 ms per step (tools/ab.sh; holding one or three more back: +0.2...0.3)
        self.defer_wgrad = not _os.environ.get("FACEOFF_NO_DEFER_WGRAD") and not self.bf16     # (the hook is a Winograd-domain GEMM launch: fp32 engine only)
        if self.bf16:
            self.winograd = self.w42 = self.fused_resblock = False
        self._pending_wgrad = {}
        self._keepalive = []
        self._cur_S = None            # state dict of the forward / backward in flight (kept Winograd planes live in it)
        if state_dict is not None:
            self.load_state_dict(state_dict)
        self.grad_ready_hook = None   # callable(layer_name) fired as soon as a layer's grads are enqueued
        self.vq_allreduce = None      # callable(stats tensor) -> summed over ranks (Quantize :63-64)

    # ------------------------------------------------------------------ state
    def set_stream_overlap(self, on: bool) -> None:
        """Run the filter-gradient kernels and the bottom Conv3d chain on their side streams (default) or, with
        ``on=False``, everything in program order on the current stream (each kernel then has the GPU to itself --
        what per-kernel timing needs)."""
        torch.cuda.synchronize(self.device)
        self.wgrad_stream, self.aux_stream, self.pack_stream, self.vq_stream = self._streams if on else (None, None, None, None)

    def load_state_dict(self, sd):
        for k, v in sd.items():
            t = torch.as_tensor(v, dtype=torch.float32).to(self.device)
            if k in self.params:
                self.params[k].copy_(t)
            elif k in self.buffers:
                self.buffers[k].copy_(t)
            else:
                raise KeyError(k)

    def state_dict(self):
        out = OrderedDict()
        for name, kind, shape in vqvae_param_specs(in_channel=self.in_channel):
            if kind == "vq":
                for s in ("embed", "cluster_size", "embed_avg"):
                    out[f"{name}.{s}"] = self.buffers[f"{name}.{s}"]
            else:
                out[name + ".weight"] = self.params[name + ".weight"]
                out[name + ".bias"] = self.params[name + ".bias"]
        return out

    # ------------------------------------------------------------------ helpers
    def _new(self, n, h, w, c, dtype=None):
        return torch.empty((n, h, w, c), device=self.device, dtype=dtype or self.act_dtype)

    def _resblock_fwd(self, prefix, x, out, out_relu):
        """ResBlock.forward (:97-101): h = relu(conv3x3(relu(x))); out = conv1x1(h) + x  [optionally relu'd]."""
        L = self.layers
        n, h, w, _ = x.shape
        hbuf = self._new(n, h, w, 32)
        c1, c3 = L[prefix + ".conv.1"], L[prefix + ".conv.3"]
        if self.fused_resblock and x.shape[-1] == 128:        # one launch: the hidden tile never leaves the CU between the convs
            ops.resblock_fwd(x, c1.wp, c1.b, c3.wp, c3.b, hbuf, out, out_relu)
            return hbuf
        c1.fwd(x, hbuf, flags=FO_IN_RELU | FO_OUT_RELU)
        c3.fwd(hbuf, out, flags=FO_OUT_RELU if out_relu else 0, add=x)
        return hbuf

    def _resblock_bwd(self, prefix, g_out, x, hbuf, g_x):
        """g_out: grad wrt (pre-ReLU) block output.  g_x = g_out + dgrad3x3(dgrad1x1(g_out)*(h>0))*(x>0)."""
        L = self.layers
        c3, c1 = L[prefix + ".conv.3"], L[prefix + ".conv.1"]
        g_h = torch.empty_like(hbuf)
        fused = self.fused_resblock_bwd and not self.bf16 and g_out.shape[-1] == 128
        if fused and max(g_out.numel(), 4 * hbuf.numel()) * 4 + 512 >= 2 ** 31:
            # fo_resblock_bwd_conv3 addresses g_out / h / g_h with 32-bit buffer offsets (FO_E_SHAPE from 2 GiB up): three launches instead
            ops._log_once(("resblock_bwd_conv3", tuple(g_out.shape)), f"faceoff_amd: resblock backward on {tuple(g_out.shape)}: tensors of 2 GiB "
                          "and more take the three-launch path (1x1 filter gradient, data gradient) instead of fo_resblock_bwd_conv3")
            fused = False
        if fused:
            # the 1x1's data, filter and bias gradients in one pass over g_out (three launches streamed it three times, each at the HBM roof)
            ops.resblock_bwd_conv3(g_out, hbuf, c3.wp, g_h, c3.gw, c3.gb)
            self._ready(c3.name)
        else:
            c3.wgrad(hbuf, g_out)
            c3.dgrad(g_out, g_h, mask=hbuf)
        c1.wgrad(x, g_h, in_relu=True)
        c1.dgrad(g_h, g_x, mask=x, add=g_out)

    def _flush_wgrad(self, everything=False):
        """Launch the deferred filter gradients that were queued from the current stream (all streams: everything=True) on the
        filter-gradient stream, behind an event recorded now."""
        cur = torch.cuda.current_stream()
        keys = list(self._pending_wgrad) if everything else [cur.cuda_stream]
        items = [it for k in keys for it in self._pending_wgrad.pop(k, [])]
        if not items:
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        self.wgrad_stream.wait_event(ev)
        hook, ops.AFTER_GEMM = ops.AFTER_GEMM, None          # (the filter-gradient kernels' own GEMMs must not re-enter)
        with torch.cuda.stream(self.wgrad_stream):
            for layer, x, g, T, in_relu in items:
                layer._wgrad(x, g, T, in_relu)
                self._ready(layer.name)
        ops.AFTER_GEMM = hook

    def _ready(self, name):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(name)

    # ------------------------------------------------------------------ forward (staged so the reference's
    # only_encode / encode_quantized / decode entry points can run the same launches)
    _PACK_FIRST = ("enc_b.blocks.0", "enc_b.blocks.2")

    def pack_filters(self, defer=False):
        """Every layer's packed filters for this step (the optimiser rewrote the checkpoint-layout weights).  With a pack stream:
        the first two layers' filters, an event, the rest, a second event; with defer=True (forward()) the caller's stream waits
        for the first before enc_b.blocks.0 and for the second before enc_b.blocks.4 (stage_encode), otherwise right here."""
        self._pack_events = None
        # Nothing to do when the checkpoint-layout weights have not changed since the last pack: a forward that follows a forward (validation,
        # the GAN loop's discriminator iterations -- the generator only moves on every other one) re-used to repack ~80 filters per call.
        # Who can write the arena: in-place torch ops (they bump the version counter the parameter views share with it) and raw-pointer kernels
        # (fo_adam_flat: FlatAdam.step calls mark_params_dirty()).  FACEOFF_ALWAYS_PACK=1 switches the check off.
        ver = self.flat_params._version
        if not self._packs_stale and self._packed_version == ver and not _os.environ.get("FACEOFF_ALWAYS_PACK"):
            return
        if _os.environ.get("FACEOFF_DIAG_NO_REPACK") and self._packed_version != -1:
            return            # DIAGNOSTIC (stale filters, timing only): what the ~130 pack / convert launches of a step cost the timed step
        self._packs_stale, self._packed_version = False, ver
        ps = self.pack_stream
        if ps is None:
            for layer in self.layers.values():
                layer.pack()
            return
        ps.wait_stream(torch.cuda.current_stream())        # the weights of this step (previous Adam launch) are final
        with torch.cuda.stream(ps):
            for name in self._PACK_FIRST:
                self.layers[name].pack()
            early = torch.cuda.Event()
            early.record(ps)
            for name, layer in self.layers.items():
                if name not in self._PACK_FIRST:
                    layer.pack()
            late = torch.cuda.Event()
            late.record(ps)
        self._pack_events = [early, late]
        if not defer:
            self._await_pack(1)
            self._pack_events = None

    def mark_params_dirty(self):
        """The parameter arena was written through a raw pointer (the Adam launch): the packed filters are stale."""
        self._packs_stale = True

    def _await_pack(self, which):
        if self._pack_events is not None and self._pack_events[which] is not None:
            torch.cuda.current_stream().wait_event(self._pack_events[which])
            self._pack_events[which] = None

    def stage_encode(self, S):
        """only_encode (:237-241): enc_b = Encoder(stride 4), enc_t = Encoder(stride 2)."""
        self._cur_S = S
        L, x8 = self.layers, S["x8"]
        N, H, W, _ = x8.shape
        h2, w2, h4, w4, h8, w8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        self._await_pack(0)
        a0 = self._new(N, h2, w2, 64); L["enc_b.blocks.0"].fwd(x8, a0, flags=FO_OUT_RELU)
        a1 = self._new(N, h4, w4, 128); L["enc_b.blocks.2"].fwd(a0, a1, flags=FO_OUT_RELU)
        self._await_pack(1)
        a2 = self._new(N, h4, w4, 128); L["enc_b.blocks.4"].fwd(a1, a2)
        a3 = self._new(N, h4, w4, 128); S["h_eb5"] = self._resblock_fwd("enc_b.blocks.5", a2, a3, False)
        eb = self._new(N, h4, w4, 128); S["h_eb6"] = self._resblock_fwd("enc_b.blocks.6", a3, eb, True)
        t0 = self._new(N, h8, w8, 64); L["enc_t.blocks.0"].fwd(eb, t0, flags=FO_OUT_RELU)
        t1 = self._new(N, h8, w8, 128); L["enc_t.blocks.2"].fwd(t0, t1)
        t2 = self._new(N, h8, w8, 128); S["h_et3"] = self._resblock_fwd("enc_t.blocks.3", t1, t2, False)
        et = self._new(N, h8, w8, 128); S["h_et4"] = self._resblock_fwd("enc_t.blocks.4", t2, et, True)
        S.update(a0=a0, a1=a1, a2=a2, a3=a3, eb=eb, t0=t0, t1=t1, t2=t2, et=et)

    def stage_conv3d(self, S):
        """Conv3dLatentPostnet x2 (:172-176,247-251).  Bottom output lands in cat_b[..., 64:192]."""
        self._cur_S = S
        L, T, eb, et = self.layers, S["T"], S["eb"], S["et"]
        N, h4, w4, _ = eb.shape
        _, h8, w8, _ = et.shape
        cat_b = self._new(N, h4, w4, 192)
        c1 = self._new(N, h4, w4, 128)
        c2 = self._new(N, h4, w4, 128)

        def bottom():
            L["conv3d_encoded_b.conv3d.0.0"].fwd(eb, c1, T=T, flags=FO_OUT_RELU)
            L["conv3d_encoded_b.conv3d.1.0"].fwd(c1, c2, T=T, flags=FO_OUT_RELU)
            L["conv3d_encoded_b.conv3d.2.0"].fwd(c2, cat_b[..., 64:192], T=T)

        if self.aux_stream is not None:      # joined in stage_quantize, right before quantize_conv_b reads cat_b
            self.aux_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.aux_stream):
                bottom()
                S["_join_conv3d_b"] = torch.cuda.Event()
                S["_join_conv3d_b"].record(self.aux_stream)
        else:
            bottom()
        d1 = self._new(N, h8, w8, 128); L["conv3d_encoded_t.conv3d.0.0"].fwd(et, d1, T=T, flags=FO_OUT_RELU)
        d2 = self._new(N, h8, w8, 128); L["conv3d_encoded_t.conv3d.1.0"].fwd(d1, d2, T=T, flags=FO_OUT_RELU)
        d3 = self._new(N, h8, w8, 128); L["conv3d_encoded_t.conv3d.2.0"].fwd(d2, d3, T=T)
        S.update(c1=c1, c2=c2, d1=d1, d2=d2, d3=d3, cat_b=cat_b)

    def stage_quantize(self, S, training, join=True):
        """encode_quantized (:261-278).  Needs S[d3] and S[cat_b][..., 64:192].  join=False (forward()): the codebook update is still running
        on its side stream when this returns; forward() joins it behind the decoder."""
        self._cur_S = S
        L, d3, cat_b = self.layers, S["d3"], S["cat_b"]
        N, h8, w8, _ = d3.shape
        _, h4, w4, _ = cat_b.shape
        f32 = torch.float32        # the quantisers' inputs stay fp32 in either engine (VQ distances, arg-min and commitment loss are fp32)
        qt_in = self._new(N, h8, w8, 64, f32); L["quantize_conv_t"].fwd(d3, qt_in)
        quant_t = self._new(N, h8, w8, 64)
        force = S.get("_force_ids") or (None, None)          # teacher-forced codes (parity tests: forward(force_ids=...))
        id_t, stats_t, S["quant_t_f32"] = self._quantize("quantize_t", qt_in, quant_t, training, force[0])
        u0 = self._new(N, h8, w8, 128); L["dec_t.blocks.0"].fwd(quant_t, u0)
        u1 = self._new(N, h8, w8, 128); S["h_dt1"] = self._resblock_fwd("dec_t.blocks.1", u0, u1, False)
        u2 = self._new(N, h8, w8, 128); S["h_dt2"] = self._resblock_fwd("dec_t.blocks.2", u1, u2, True)
        L["dec_t.blocks.4"].fwd(u2, cat_b[..., 0:64])                      # torch.cat([dec_t, enc_b], 1) :271
        if S.get("_join_conv3d_b") is not None:
            torch.cuda.current_stream().wait_event(S.pop("_join_conv3d_b"))
        qb_in = self._new(N, h4, w4, 64, f32); L["quantize_conv_b"].fwd(cat_b, qb_in)
        cat_d = self._new(N, h4, w4, 128)
        id_b, stats_b, S["quant_b_f32"] = self._quantize("quantize_b", qb_in, cat_d[..., 64:128], training, force[1])
        S.update(qt_in=qt_in, quant_t=quant_t, u0=u0, u1=u1, u2=u2, qb_in=qb_in, cat_d=cat_d, id_t=id_t, id_b=id_b)
        S["_vq_stats"] = (stats_t, stats_b)      # read by the codebook update on its side stream: alive until S goes (after the join)
        # diff = diff_t + diff_b, each mean((q - x)^2) (:77,268,276,278)
        S["diff"] = stats_t[0:1] / float(qt_in.numel()) + stats_b[0:1] / float(qb_in.numel())
        # EMA codebook update after the (optional) cross-rank sum of the statistics (:59-75)
        if training:
            import contextlib
            with (torch.cuda.stream(self.vq_stream) if self.vq_stream is not None else contextlib.nullcontext()):
                for lvl, st in (("t", stats_t), ("b", stats_b)):
                    if self.vq_allreduce is not None:
                        self.vq_allreduce(st[1:])
                    ops.vq_ema(self.buffers[f"quantize_{lvl}.embed"], self.buffers[f"quantize_{lvl}.cluster_size"],
                               self.buffers[f"quantize_{lvl}.embed_avg"], st)
                if self.vq_stream is not None:
                    self._vq_done = torch.cuda.Event()
                    self._vq_done.record(self.vq_stream)
            if join and self.vq_stream is not None:
                torch.cuda.current_stream().wait_stream(self.vq_stream)

    def stage_decode(self, S):
        """decode (:280-285).  Needs S[quant_t] and S[cat_d][..., 64:128] (= quant_b)."""
        self._cur_S = S
        L, quant_t, cat_d = self.layers, S["quant_t"], S["cat_d"]
        N, h4, w4, _ = cat_d.shape
        L["upsample_t"].fwd(quant_t, cat_d[..., 0:64])                     # torch.cat([upsample_t, quant_b], 1) :282
        v0 = self._new(N, h4, w4, 128); L["dec.blocks.0"].fwd(cat_d, v0)
        v1 = self._new(N, h4, w4, 128); S["h_d1"] = self._resblock_fwd("dec.blocks.1", v0, v1, False)
        v2 = self._new(N, h4, w4, 128); S["h_d2"] = self._resblock_fwd("dec.blocks.2", v1, v2, True)
        w1 = self._new(N, 2 * h4, 2 * w4, 64); L["dec.blocks.4"].fwd(v2, w1, flags=FO_OUT_RELU)
        dec = torch.empty((N, 4 * h4, 4 * w4, 8), device=self.device); L["dec.blocks.6"].fwd(w1, dec)   # (fp32 in either engine; every pixel, all 8 floats, is written)
        S.update(v0=v0, v1=v1, v2=v2, w1=w1, dec=dec)

    def forward(self, img_nchw, training=True, T=None, force_ids=None):
        """VQVAE.forward (:243-259).  img_nchw [N,6,H,W] (N = B*T frames), or the pair (source, background) of
        [N,3,H,W] tensors process_data would concatenate (utils.py:32) -- the cat then happens inside the layout kernel.
        Returns S: dict of saved activations incl. S[dec] NHWC [N,H,W,8], S[diff] [1], S[id_t], S[id_b].
        force_ids = (id_t [N,H/8,W/8], id_b [N,H/4,W/4]): teacher-forced codes -- both quantisers skip the search and use these
        (gather, straight-through value, commitment loss and EMA statistics all on the given codes); everything else is the step."""
        parts = None
        if isinstance(img_nchw, (tuple, list)):
            parts = img_nchw
            N, _, H, W = parts[0].shape
            Cin = parts[0].shape[1] + parts[1].shape[1]
        else:
            N, Cin, H, W = img_nchw.shape
        T = T or self.clip_len or N
        assert N % T == 0, f"N={N} frames is not a whole number of clips of T={T}"
        assert H % 8 == 0 and W % 8 == 0, "spatial size must be a multiple of 8"
        self.pack_filters(defer=True)
        self.keep_wino_v = bool(training)
        assert Cin <= 8
        if self.bf16:
            x8 = ops.cat_nchw_to_nhwc8_bf16(*parts) if parts is not None else ops.cat_nchw_to_nhwc8_bf16(img_nchw)
        else:
            x8 = ops.cat_nchw_to_nhwc8(*parts) if parts is not None else ops.nchw_to_nhwc(img_nchw, cpad=ops.pad_in(Cin))
        S = {"T": T, "x8": x8}
        if force_ids is not None:
            S["_force_ids"] = tuple(force_ids)
        self.stage_encode(S)
        self.stage_conv3d(S)
        self.stage_quantize(S, training, join=False)
        self.stage_decode(S)
        if self.vq_stream is not None and training:      # the codebook update ran beside the decoder; whoever reads the buffers next is behind it
            torch.cuda.current_stream().wait_stream(self.vq_stream)
        return S

    def diag_queue_shift(self):
        """Diagnostics: FACEOFF_DIAG_QUEUE_SHIFT=n throw-away streams take the next hardware queues first (what a communicator's or another engine's would do)."""
        if _os.environ.get("FACEOFF_DIAG_QUEUE_SHIFT") and self.device.type == "cuda" and not getattr(self, "_shift", None):
            self._shift = [torch.cuda.Stream(device=self.device) for _ in range(int(_os.environ["FACEOFF_DIAG_QUEUE_SHIFT"]))]
            for st in self._shift:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)

    def keep_wgrad_off_main_queue(self):
        """HIP gives a stream its hardware queue at first use, from a pool of GPU_MAX_HW_QUEUES = 4; in order of first use the queues go 2, 3, 4, 4, 3, 2, 1, 4, 3 ..
        (queue 1 is the null stream's).  Which queue the filter-gradient stream gets therefore depends on how many streams the PROCESS has used before this engine's
        -- another engine's, a communicator's -- and when it is queue 1 its launches line up behind the data-gradient chain they are meant to run beside: config 2
        39.7 -> 41.2 ms, config 3 32.3 -> 34.0, config 5 13.2 -> 13.6 (three throw-away streams in front).  Checked here, once, by the trainers at
        construction: a small launch on the stream must finish while long launches occupy the current stream; if it does not, the next pooled stream takes its
        place (at most 8 tries).  ~2 ms and a transient 512 MB.  FACEOFF_NO_QUEUE_CHECK=1: as it comes."""
        if self.device.type != "cuda" or self.wgrad_stream is None:
            return
        self.diag_queue_shift()
        if _os.environ.get("FACEOFF_NO_QUEUE_CHECK"):
            return
        for st in (self.pack_stream, self.aux_stream, self.vq_stream, self.wgrad_stream):       # (the order a step reaches them in)
            if st is not None:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)
        for _ in range(8):
            if _runs_beside_current(self.wgrad_stream, self.device):
                break
            self.wgrad_stream = torch.cuda.Stream(device=self.device)
        self._streams = (self.wgrad_stream,) + tuple(self._streams[1:])       # (set_stream_overlap(True) restores from this tuple)

    def _quantize(self, name, x, q_out, training, force_ind=None):
        # The previous step's codebook update: long finished, and every path that launches it joins the side stream before it returns (stage_quantize /
        # forward) -- what is awaited here is the EVENT recorded behind that update, not the stream: wait_stream() would drop a fresh marker into
        # vq_stream's hardware queue, which HIP shares with other streams (six streams, four queues by default), and the main stream then sat behind
        # whatever the neighbour had queued: 0.86 ms of config 3's forward behind the bottom Conv3d chain (tools/stream_timeline.py, round 6).
        if self.vq_stream is not None and self._vq_done is not None and not _os.environ.get("FACEOFF_VQ_WAIT_STREAM"):
            torch.cuda.current_stream().wait_event(self._vq_done)
        elif self.vq_stream is not None:
            torch.cuda.current_stream().wait_stream(self.vq_stream)
        embedT, enorm = ops.vq_prepare(self.buffers[name + ".embed"])
        stats = torch.zeros(1 + 512 + 512 * 64, device=self.device)
        if self.bf16:        # the straight-through output twice: fp32 (kept for the backward's 2 (x - q) / numel term), bf16 for the next conv
            q32 = torch.empty(x.shape, device=self.device, dtype=torch.float32)
            ind = ops.vq_assign_bf16out(x, embedT, enorm, q32, q_out, stats, training, stats_stream=self.vq_stream, force_ind=force_ind)
            return ind, stats, q32
        ind = ops.vq_assign(x, embedT, enorm, q_out, stats, training, stats_stream=self.vq_stream, force_ind=force_ind)
        return ind, stats, None

    def _vq_bwd(self, gq, x, q, q32, g_diff):
        """Quantize backward (:77-78): g_x = g_q + g_diff * 2 (x - q) / numel; returns g_x in the engine's activation dtype."""
        gx = torch.empty(x.shape, device=self.device, dtype=self.act_dtype)
        if self.bf16:
            ops.vq_bwd_bf16(gq, x, q32, g_diff, gx)
        else:
            ops.vq_bwd(gq, x, q, g_diff, gx)
        return gx

    # ------------------------------------------------------------------ backward
    def backward(self, S, g_dec, g_diff):
        """S: the dict forward() returned.  g_dec NHWC [N,H,W,8] grad wrt dec; g_diff float32[1] device tensor.
        Fills self.grads (flat arena).  Order = reverse forward, so arena slices complete back-to-front."""
        self._pending_wgrad.clear()        # (nothing may survive an aborted backward)
        if self.defer_wgrad and self.wgrad_stream is not None:
            ops.AFTER_GEMM = self._flush_wgrad
        try:
            self._backward(S, g_dec, g_diff)
        finally:
            # also after an exception: a hook left installed would launch this backward's stale filter gradients from the
            # next forward's Winograd GEMMs (into the gradient arena, popping from a dead S, firing the DDP hook out of turn)
            ops.AFTER_GEMM = None
            self._pending_wgrad.clear()
            self._keepalive.clear()
            S.pop("_wino_v", None)
            self._cur_S = None

    def _backward(self, S, g_dec, g_diff):
        L = self.layers
        T = S["T"]
        self._cur_S = S
        new_like = torch.empty_like
        if self.bf16 and g_dec.dtype != torch.bfloat16:      # the loss kernels hand over an fp32 gradient: stored as bf16 once, here
            g_dec = ops.to_bf16(g_dec)
        # ---- dec (Decoder stride 4)
        l6, l4 = L["dec.blocks.6"], L["dec.blocks.4"]
        l6.wgrad(S["w1"], g_dec)
        g_w1 = new_like(S["w1"]); l6.dgrad(g_dec, g_w1, mask=S["w1"])
        l4.wgrad(S["v2"], g_w1)
        g_v2 = new_like(S["v2"]); l4.dgrad(g_w1, g_v2, mask=S["v2"])
        g_v1 = new_like(S["v1"]); self._resblock_bwd("dec.blocks.2", g_v2, S["v1"], S["h_d2"], g_v1)
        g_v0 = new_like(S["v0"]); self._resblock_bwd("dec.blocks.1", g_v1, S["v0"], S["h_d1"], g_v0)
        l0 = L["dec.blocks.0"]
        l0.wgrad(S["cat_d"], g_v0)
        g_cat_d = new_like(S["cat_d"]); l0.dgrad(g_v0, g_cat_d)
        # ---- upsample_t
        up = L["upsample_t"]
        up.wgrad(S["quant_t"], g_cat_d[..., 0:64])
        g_quant_t = new_like(S["quant_t"]); up.dgrad(g_cat_d[..., 0:64], g_quant_t)
        # ---- quantize_b (straight-through + commitment) and quantize_conv_b
        g_qb_in = self._vq_bwd(g_cat_d[..., 64:128], S["qb_in"], S["cat_d"][..., 64:128], S.get("quant_b_f32"), g_diff)
        qcb = L["quantize_conv_b"]
        qcb.wgrad(S["cat_b"], g_qb_in)
        g_cat_b = new_like(S["cat_b"]); qcb.dgrad(g_qb_in, g_cat_b)
        # ---- conv3d_encoded_b, all but its last dgrad (its output gradient is cat_b[..., 64:192]): on the second
        # side stream beside the top-level chain below; joined where the two gradients of enc_b meet
        kb2, kb1, kb0 = (L[f"conv3d_encoded_b.conv3d.{i}.0"] for i in (2, 1, 0))
        g_c3 = g_cat_b[..., 64:192]
        g_c2, g_c1 = new_like(S["c2"]), new_like(S["c1"])

        def bottom3d():
            kb2.wgrad(S["c2"], g_c3, T=T)
            kb2.dgrad(g_c3, g_c2, T=T, mask=S["c2"])
            kb1.wgrad(S["c1"], g_c2, T=T)
            kb1.dgrad(g_c2, g_c1, T=T, mask=S["c1"])
            kb0.wgrad(S["eb"], g_c1, T=T)
            if self.defer_wgrad and self.wgrad_stream is not None:
                self._flush_wgrad()       # no GEMM follows on this stream: the chain's last filter gradient must not wait for the end

        if self.aux_stream is not None:
            self.aux_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.aux_stream):
                bottom3d()
        # ---- dec_t
        dt4 = L["dec_t.blocks.4"]
        dt4.wgrad(S["u2"], g_cat_b[..., 0:64])
        g_u2 = new_like(S["u2"]); dt4.dgrad(g_cat_b[..., 0:64], g_u2, mask=S["u2"])
        g_u1 = new_like(S["u1"]); self._resblock_bwd("dec_t.blocks.2", g_u2, S["u1"], S["h_dt2"], g_u1)
        g_u0 = new_like(S["u0"]); self._resblock_bwd("dec_t.blocks.1", g_u1, S["u0"], S["h_dt1"], g_u0)
        dt0 = L["dec_t.blocks.0"]
        dt0.wgrad(S["quant_t"], g_u0)
        g_quant_t2 = new_like(S["quant_t"]); dt0.dgrad(g_u0, g_quant_t2, add=g_quant_t)   # fan-in of quant_t's two uses
        # ---- quantize_t and quantize_conv_t
        g_qt_in = self._vq_bwd(g_quant_t2, S["qt_in"], S["quant_t"], S.get("quant_t_f32"), g_diff)
        qct = L["quantize_conv_t"]
        qct.wgrad(S["d3"], g_qt_in)
        g_d3 = new_like(S["d3"]); qct.dgrad(g_qt_in, g_d3)
        # ---- conv3d_encoded_t
        k2, k1, k0 = (L[f"conv3d_encoded_t.conv3d.{i}.0"] for i in (2, 1, 0))
        k2.wgrad(S["d2"], g_d3, T=T)
        g_d2 = new_like(S["d2"]); k2.dgrad(g_d3, g_d2, T=T, mask=S["d2"])
        k1.wgrad(S["d1"], g_d2, T=T)
        g_d1 = new_like(S["d1"]); k1.dgrad(g_d2, g_d1, T=T, mask=S["d1"])
        k0.wgrad(S["et"], g_d1, T=T)
        g_t3 = new_like(S["et"]); k0.dgrad(g_d1, g_t3, T=T, mask=S["et"])     # mask = encoder's final ReLU
        # ---- enc_t
        g_t2 = new_like(S["t2"]); self._resblock_bwd("enc_t.blocks.4", g_t3, S["t2"], S["h_et4"], g_t2)
        g_t1 = new_like(S["t1"]); self._resblock_bwd("enc_t.blocks.3", g_t2, S["t1"], S["h_et3"], g_t1)
        e2, e0 = L["enc_t.blocks.2"], L["enc_t.blocks.0"]
        e2.wgrad(S["t0"], g_t1)
        g_t0 = new_like(S["t0"]); e2.dgrad(g_t1, g_t0, mask=S["t0"])
        e0.wgrad(S["eb"], g_t0)
        g_eb_t = new_like(S["eb"]); e0.dgrad(g_t0, g_eb_t, mask=S["eb"])      # grad via enc_t, through enc_b's final ReLU
        # ---- join the bottom Conv3d chain; its last dgrad fans in the gradient that came through enc_t
        if self.aux_stream is not None:
            torch.cuda.current_stream().wait_stream(self.aux_stream)
        else:
            bottom3d()
        g_a4 = new_like(S["eb"]); kb0.dgrad(g_c1, g_a4, T=T, mask=S["eb"], add=g_eb_t)
        # ---- enc_b
        g_a3 = new_like(S["a3"]); self._resblock_bwd("enc_b.blocks.6", g_a4, S["a3"], S["h_eb6"], g_a3)
        g_a2 = new_like(S["a2"]); self._resblock_bwd("enc_b.blocks.5", g_a3, S["a2"], S["h_eb5"], g_a2)
        b4, b2, b0 = L["enc_b.blocks.4"], L["enc_b.blocks.2"], L["enc_b.blocks.0"]
        # The last layers' filter gradients on the CALLER's stream, behind the last data gradient: the side stream reaches the end of the backward
        # with a backlog (bf16 engine at config 3: 1.15 ms of filter gradients and their reduce launches after the main stream's last kernel;
        # fp32 engine: 1.85 ms; tools/timeline.py, per-queue ends), and the main stream has nothing left to do but wait for it.  Two layers
        # (FACEOFF_TAIL_WGRADS): same-device A/B config 2 39.67 -> 39.57 ms, config 3 34.84 -> 34.48; three puts the main stream 1.6 ms behind.
        n_tail = self.tail_wgrads if self.wgrad_stream is not None else 0
        tail = []

        def wgrad_or_tail(layer, x, g, rank):
            if rank < n_tail:
                tail.append((layer, x, g))
            else:
                layer.wgrad(x, g)
        wgrad_or_tail(b4, S["a1"], g_a2, 2)
        g_a1 = new_like(S["a1"]); b4.dgrad(g_a2, g_a1, mask=S["a1"])
        wgrad_or_tail(b2, S["a0"], g_a1, 1)
        g_a0 = new_like(S["a0"]); b2.dgrad(g_a1, g_a0, mask=S["a0"])
        wgrad_or_tail(b0, S["x8"], g_a0, 0)
        for layer, x, g in reversed(tail):               # (newest gradient first: its operands are the hottest in L2)
            layer._wgrad(x, g, 1, False)
            self._ready(layer.name)
        ops.AFTER_GEMM = None
        if self.wgrad_stream is not None and self._pending_wgrad:
            self._flush_wgrad(everything=True)
        if self.wgrad_stream is not None:          # join: every filter gradient is in the arena after this
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    # ------------------------------------------------------------------ fused train step (bench / trainer fast path)
    def loss_and_backward(self, img_nchw, gt_nchw, T=None, latent_weight=1.0, force_ids=None):
        """run_step + backward (train_faceoff_perceptual.py:32-47,98-100) for recon + latent loss.
        Returns device scalars (recon, latent).  Gradients land in self.flat_grads.  force_ids: see forward()."""
        S = self.forward(img_nchw, training=True, T=T, force_ids=force_ids)
        dec = S["dec"]
        acc = torch.zeros(1, device=self.device)
        one = torch.ones(1, device=self.device)
        g_dec = torch.empty_like(dec)
        ops.mse_slice_fwd_bwd(dec, gt_nchw, acc, one, g_dec)       # loss value and gradient in one pass over dec and gt
        recon = acc / float(gt_nchw.numel())
        self.backward(S, g_dec, one * latent_weight)
        return recon, S["diff"], S
