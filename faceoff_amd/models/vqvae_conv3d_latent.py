"""Drop-in mirror of the reference `models/vqvae_conv3d_latent.py` (VQVAE :192-295, Quantize :33-83)
whose arithmetic runs in hand-written gfx950 kernels (libfaceoff_hip.so) instead of torch.nn / cuDNN.

Kept from the reference: constructor signature, `forward(input) -> (dec, diff[1])`,
`only_encode`, `encode_quantized`, `decode`, `decode_code`, nn.Module semantics (`.train()/.eval()`
toggles the EMA branch, `.parameters()` feeds any optimiser, `.to(device)`), and `state_dict` keys /
shapes (70 parameters + 6 buffers, NCHW/OIHW) so reference checkpoints load unchanged.

Different by design: the whole forward is ONE autograd node (the engine keeps exactly the
activations its explicit backward needs); parameters live in one flat arena, gradients in another
(data-parallel buckets are slices of it); and the clip axis is explicit: `forward` takes
[N,6,H,W] (N frames = one clip, the literal reference) or [B,T,6,H,W] / `clip_len=T`.

The HIP library is mandatory: construction on a CUDA/HIP device fails loudly if it is missing.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops
from ..engine import VQVAEEngine
from ..synth import vqvae_param_specs
from .. import distributed as dist_fn


class _Holder(nn.Module):
    """Container that reproduces the reference's nested module names (enc_b.blocks.0.weight ...)."""


def _register(root: nn.Module, dotted: str, tensor, buffer=False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class _VQVAEFunction(torch.autograd.Function):
    """forward + backward of the whole network as one node; parameters are passed so autograd routes
    their gradients (which the engine writes into its flat gradient arena)."""

    @staticmethod
    def forward(ctx, model, img, T, *params):
        eng = model._engine
        # this module's nn.Parameters are `.data` aliases of the engine's arena (their own version counters): whoever steps or loads them --
        # torch.optim, load_state_dict, DDP's broadcast of module states -- writes around the counter engine.pack_filters() watches, so a
        # forward through the MODULE always repacks its filters (the engine-level trainers, which own their optimiser, skip unchanged weights)
        eng.mark_params_dirty()
        S = eng.forward(img, training=model.training, T=T)
        ctx.model, ctx.S = model, S
        dec = ops.nhwc_to_nchw(S["dec"], model.in_channel)
        ctx.mark_non_differentiable(S["id_t"], S["id_b"])
        return dec, S["diff"], S["id_t"], S["id_b"]

    @staticmethod
    def backward(ctx, g_dec, g_diff, *_):
        model, S = ctx.model, ctx.S
        eng = model._engine
        N, H, W = S["dec"].shape[:3]
        if g_dec is None:
            g8 = torch.zeros((N, H, W, 8), device=eng.device)
        else:
            g8 = ops.nchw_to_nhwc(g_dec.contiguous(), cpad=8)
        if g_diff is None:
            g_diff = torch.zeros(1, device=eng.device)
        eng.backward(S, g8, g_diff.contiguous().reshape(1).float())
        ctx.S = None
        # the arena is overwritten by the next backward: hand autograd views of ONE flat copy (one launch, not 70)
        flat = eng.flat_grads.clone()
        grads = tuple(flat[o:o + n].view(eng.params[k].shape) for k, (o, n) in ((k, eng.offsets[k]) for k in model._param_keys))
        return (None, None, None) + grads


class Quantize(nn.Module):
    """Reference Quantize(dim, n_embed, decay, eps) (:33-83) on the fused VQ kernels (dim 64, 512 codes)."""

    def __init__(self, dim, n_embed, decay=0.99, eps=1e-5):
        super().__init__()
        if dim != 64 or n_embed != 512:
            raise ValueError("the gfx950 VQ kernel is built for dim=64, n_embed=512 (the FaceOff configuration)")
        self.dim, self.n_embed, self.decay, self.eps = dim, n_embed, decay, eps
        embed = torch.randn(dim, n_embed)
        self.register_buffer("embed", embed)
        self.register_buffer("cluster_size", torch.zeros(n_embed))
        self.register_buffer("embed_avg", embed.clone())

    def forward(self, input):
        x = input.contiguous().float()
        embedT, enorm = ops.vq_prepare(self.embed)
        q = torch.empty_like(x)
        stats = torch.zeros(1 + 512 + 512 * 64, device=x.device)
        ind = ops.vq_assign(x.view(-1, 1, 1, 64) if x.dim() != 4 else x, embedT, enorm,
                            q.view(-1, 1, 1, 64) if q.dim() != 4 else q, stats, self.training).view(x.shape[:-1])
        if self.training:
            dist_fn.all_reduce(stats[1:])                      # :63-64, one message
            ops.vq_ema(self.embed, self.cluster_size, self.embed_avg, stats, self.decay, self.eps)
        diff = stats[0:1] / float(x.numel())
        quantize, diff = _QuantizeSTE.apply(input, q, diff)
        return quantize, diff.reshape(()), ind

    def embed_code(self, embed_id):
        """:82-83 F.embedding(embed_id, embed^T) -> [..., dim], by the gather kernel (fo_vq_gather)."""
        embedT, _ = ops.vq_prepare(self.embed)
        q = torch.empty((*embed_id.shape, self.dim), device=self.embed.device, dtype=torch.float32)
        ops.vq_gather(embed_id.contiguous(), embedT, q.view(-1, 1, 1, self.dim))
        return q


class _QuantizeSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, q, diff):
        ctx.save_for_backward(x, q)
        return q.view_as(x), diff

    @staticmethod
    def backward(ctx, gq, gdiff):
        x, q = ctx.saved_tensors
        xc = x.contiguous().float()
        gx = torch.empty_like(xc)
        v4 = lambda t: t.contiguous().view(-1, 1, 1, 64)
        ops.vq_bwd(v4(gq if gq is not None else torch.zeros_like(xc)), v4(xc), v4(q),
                   (gdiff if gdiff is not None else torch.zeros(1, device=x.device)).reshape(1).float().contiguous(), v4(gx))
        return gx.view_as(x), None, None


class VQVAE(nn.Module):
    def __init__(self, in_channel=3, channel=128, n_res_block=2, n_res_channel=32, embed_dim=64, n_embed=512,
                 decay=0.99, residual=False, clip_len=None):
        super().__init__()
        if (channel, n_res_block, n_res_channel, embed_dim, n_embed) != (128, 2, 32, 64, 512):
            # channel must be 128 in the reference too (hard-coded Conv3dLatentPostnet(128), :230-231)
            raise ValueError("the gfx950 engine is built for the FaceOff configuration: channel=128, n_res_block=2, "
                             "n_res_channel=32, embed_dim=64, n_embed=512")
        if in_channel > 8:
            raise ValueError("in_channel must be <= 8")
        self.in_channel = in_channel
        self.residual = residual          # stored and unused, as in the reference (:227)
        self.clip_len = clip_len
        self._param_keys = []
        for name, kind, shape in vqvae_param_specs(in_channel=in_channel):
            if kind == "vq":
                e = torch.randn(shape)                                        # Quantize.__init__ :42-45
                _register(self, name + ".embed", e, buffer=True)
                _register(self, name + ".cluster_size", torch.zeros(shape[1]), buffer=True)
                _register(self, name + ".embed_avg", e.clone(), buffer=True)
                continue
            fan_in = shape[1] * math.prod(shape[2:])                          # torch default conv init
            bound = 1.0 / math.sqrt(fan_in)
            nb = shape[1] if kind == "convT" else shape[0]
            _register(self, name + ".weight", torch.empty(shape).uniform_(-bound, bound))
            _register(self, name + ".bias", torch.empty(nb).uniform_(-bound, bound))
            self._param_keys += [name + ".weight", name + ".bias"]
        self._engine = None

    # ------------------------------------------------------------------ engine binding
    def _bind(self, device):
        """(Re)create the engine on `device` and re-point parameters / buffers at its arenas."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("faceoff_amd.VQVAE computes only on an MI355X (cuda/HIP device): there is no CPU fallback. "
                               "Move the module with .to('cuda') first.")
        named = dict(self.named_parameters())
        bufs = dict(self.named_buffers())
        eng = self._engine
        same = eng is not None and eng.device == device and all(
            named[k].data_ptr() == eng.params[k].data_ptr() for k in self._param_keys)
        if same:
            return eng
        sd = {k: v.detach() for k, v in list(named.items()) + list(bufs.items())}
        eng = VQVAEEngine(sd, device, in_channel=self.in_channel, clip_len=self.clip_len)
        for k in self._param_keys:
            named[k].data = eng.params[k]
            named[k].grad = None
        for k, b in bufs.items():
            b.data = eng.buffers[k]
        eng.vq_allreduce = dist_fn.fused_vq_allreduce()      # checks the world size at call time (a no-op for one rank)
        self._engine = eng
        return eng

    @property
    def engine(self):
        return self._bind(next(self.parameters()).device)

    def _apply(self, fn, *a, **kw):
        """`.to(device)` / `.cuda()`: once the parameters sit on a GPU, re-home them in the engine's flat arena right
        away (not lazily at the first forward), so anything that captures the parameters afterwards -- an optimiser,
        nn.parallel.DistributedDataParallel -- sees their final storage."""
        out = super()._apply(fn, *a, **kw)
        dev = next(self.parameters()).device
        if dev.type == "cuda":
            self._bind(dev)
        return out

    # ------------------------------------------------------------------ reference API
    def forward(self, input):
        """input [N,6,H,W] (one clip of N frames, or `clip_len`-frame clips) or [B,T,6,H,W].
        Returns (dec [N,6,H,W], diff [1]) like the reference (:243-259)."""
        T = self.clip_len
        if input.dim() == 5:
            T = input.shape[1]
            input = input.reshape(-1, *input.shape[2:])
        self._bind(input.device)
        params = [p for _, p in self.named_parameters()]
        dec, diff, id_t, id_b = _VQVAEFunction.apply(self, input.float().contiguous(), T, *params)
        self._last_ids = (id_t, id_b)
        return dec, diff

    # (the staged entry points hand tensors in and out as fp32 NCHW like the reference; with FACEOFF_DTYPE=bf16 the engine's
    # activations are bf16 channels-last tensors: converted at this boundary)
    @staticmethod
    def _act_in(eng, x_nhwc_f32):
        return ops.to_bf16(x_nhwc_f32) if eng.bf16 else x_nhwc_f32

    @staticmethod
    def _act_out(t, c):
        return ops.nhwc_to_nchw(ops.to_f32(t) if t.dtype == torch.bfloat16 else t, c)

    def _pack(self, eng):
        eng.mark_params_dirty()           # (see _VQVAEFunction.forward: the module never trusts the arena's version counter)
        eng.pack_filters()

    @torch.no_grad()
    def only_encode(self, input):
        """:237-241 -> (enc_b [N,128,H/4,W/4], enc_t [N,128,H/8,W/8])."""
        eng = self._bind(input.device)
        self._pack(eng)
        S = {"T": self.clip_len or input.shape[0], "x8": self._act_in(eng, ops.nchw_to_nhwc(input.float().contiguous(), cpad=8))}
        eng.stage_encode(S)
        return self._act_out(S["eb"], 128), self._act_out(S["et"], 128)

    @torch.no_grad()
    def encode_quantized(self, enc_b, enc_t):
        """:261-278 -> (quant_t, quant_b, diff, id_t, id_b).  Inference entry point (no autograd)."""
        eng = self._bind(enc_b.device)
        self._pack(eng)
        N, _, h4, w4 = enc_b.shape
        S = {"T": self.clip_len or N, "d3": self._act_in(eng, ops.nchw_to_nhwc(enc_t.float().contiguous()))}
        cat_b = torch.empty((N, h4, w4, 192), device=eng.device, dtype=eng.act_dtype)
        cat_b[..., 64:192] = self._act_in(eng, ops.nchw_to_nhwc(enc_b.float().contiguous()))
        S["cat_b"] = cat_b
        eng.stage_quantize(S, self.training)
        self._last_S = S
        quant_t = S["quant_t_f32"] if eng.bf16 else S["quant_t"]          # (the quantiser's fp32 output where the engine keeps one)
        quant_b = S["quant_b_f32"] if eng.bf16 else S["cat_d"][..., 64:128]
        return (ops.nhwc_to_nchw(quant_t, 64), ops.nhwc_to_nchw(quant_b, 64), S["diff"], S["id_t"], S["id_b"])

    @torch.no_grad()
    def decode(self, quant_t, quant_b):
        """:280-285 -> dec [N,6,H,W]."""
        eng = self._bind(quant_t.device)
        self._pack(eng)
        N, _, h4, w4 = quant_b.shape
        cat_d = torch.empty((N, h4, w4, 128), device=eng.device, dtype=eng.act_dtype)
        cat_d[..., 64:128] = self._act_in(eng, ops.nchw_to_nhwc(quant_b.float().contiguous()))
        S = {"quant_t": self._act_in(eng, ops.nchw_to_nhwc(quant_t.float().contiguous())), "cat_d": cat_d}
        eng.stage_decode(S)
        return ops.nhwc_to_nchw(S["dec"], self.in_channel)

    @torch.no_grad()
    def decode_code(self, code_t, code_b):
        """:287-295: codes -> codebook rows -> decode."""
        eng = self._bind(code_t.device)
        outs = []
        for lvl, code in (("t", code_t), ("b", code_b)):
            embedT, _ = ops.vq_prepare(eng.buffers[f"quantize_{lvl}.embed"])
            q = torch.empty((*code.shape, 64), device=eng.device)
            ops.vq_gather(code.contiguous(), embedT, q)
            outs.append(ops.nhwc_to_nchw(q, 64))
        return self.decode(outs[0], outs[1])

    # the reference loads checkpoints saved from DDP-wrapped models by stripping "module." (:178-185)
    def load_state_dict(self, state_dict, strict=True, assign=False):
        state_dict = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict)
