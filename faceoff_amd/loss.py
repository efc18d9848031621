"""Mirror of the reference `loss.VQLPIPS` (loss.py:27-33): `VQLPIPS()(targets, reconstructions)` -> 0-dim tensor,
all parameters frozen, always eval -- computed by faceoff_amd.lpips.LPIPSEngine on the gfx950 kernels.

The reference constructor downloads vgg.pth + torchvision VGG-16 (lpips.py:12-48,118): there is no network
here, so weights come from `load_state_dict` (reference key names under `perceptual_loss.`) or the
`state_dict=` argument; constructing without weights and calling forward raises.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .lpips import LPIPSEngine, conv_keys


class _LPIPSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, targets, recon):
        eng = module._bind(recon.device)
        N, _, H, W = recon.shape
        dec = ops.nchw_to_nhwc(recon.float().contiguous(), cpad=8)
        g_dec = torch.zeros_like(dec) if recon.requires_grad else None
        loss = eng.loss_and_grad(targets.float().contiguous(), dec, g_dec)
        ctx.g_dec = g_dec
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.g_dec is None:
            return None, None, None
        grad = ops.nhwc_to_nchw(ctx.g_dec, 3) * g
        return None, None, grad


class _LPIPSPerImageFunction(torch.autograd.Function):
    """`models.lpips.LPIPS.forward(input, target)` (reference models/lpips.py:80-93) -> [N,1,1,1], differentiable in both arguments like the
    reference's (the shim used to hand out a detached tensor).  The distance is symmetric in its two images, and the kernels
    produce d mean(val) / d (one image) in the same pass as the values: one pass per argument that requires a gradient, with that argument in
    the reconstruction slot.  A backward whose incoming gradient is the same for every image -- `.mean()`, `.sum()`, a scalar weight: every use
    in the reference tree -- is that tensor rescaled; any other incoming gradient is served image by image (one single-frame pass each:
    correct, not fast)."""

    @staticmethod
    def forward(ctx, module, inp, target):
        eng = module._bind(inp.device)
        imgs = (inp.float().contiguous(), target.float().contiguous())
        ctx.module, ctx.n, ctx.g_mean, ctx.nhwc = module, inp.shape[0], [None, None], [None, None]
        vals = None
        for slot in (0, 1):
            if not (inp, target)[slot].requires_grad and (slot == 1 or vals is not None or target.requires_grad):
                continue                     # (with no gradient asked for at all, slot 0 runs once for the values)
            ctx.nhwc[slot] = ops.nchw_to_nhwc(imgs[slot], cpad=8)
            need = (inp, target)[slot].requires_grad
            ctx.g_mean[slot] = torch.zeros_like(ctx.nhwc[slot]) if need else None
            eng.loss_and_grad(imgs[1 - slot], ctx.nhwc[slot], ctx.g_mean[slot])
            if vals is None:
                vals = eng.last_per_image.clone()
        ctx.other = imgs
        return vals.reshape(-1, 1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        flat = g.reshape(-1).float()
        uniform = bool((flat == flat[0]).all())
        out = [None, None]
        for slot in (0, 1):
            if ctx.g_mean[slot] is None:
                continue
            if uniform:
                out[slot] = ops.nhwc_to_nchw(ctx.g_mean[slot], 3) * (flat[0] * ctx.n)
                continue
            eng = ctx.module._bind(g.device)
            acc = torch.zeros_like(ctx.nhwc[slot])
            for n in range(ctx.n):          # d val[n] / d image[n], one frame at a time
                eng.loss_and_grad(ctx.other[1 - slot][n:n + 1], ctx.nhwc[slot][n:n + 1], acc[n:n + 1])
            out[slot] = ops.nhwc_to_nchw(acc, 3) * flat.reshape(-1, 1, 1, 1)
        return None, out[0], out[1]


class VQLPIPS(nn.Module):
    def __init__(self, state_dict=None, dtype="fp32"):
        """dtype: "fp32" (reference arithmetic) or "bf16" (BASELINE config 3: bf16 storage / MFMA operands, fp32
        accumulation -- what torch.autocast(bfloat16) around the reference's VQLPIPS would compute)."""
        super().__init__()
        assert dtype in ("fp32", "bf16")
        self.compute_dtype = dtype
        holder = nn.Module()
        for key, ci, co, _ in conv_keys():
            _register(holder, key + ".weight", torch.zeros(co, ci, 3, 3))
            _register(holder, key + ".bias", torch.zeros(co))
        for k, c in enumerate([64, 128, 256, 512, 512]):
            _register(holder, f"lin{k}.model.1.weight", torch.zeros(1, c, 1, 1))
        self.perceptual_loss = holder
        for p in self.parameters():
            p.requires_grad = False                       # lpips.py:63-64
        self._engine = None
        self._loaded = False
        if state_dict is not None:
            self.load_state_dict(state_dict)
        self.eval()

    def load_state_dict(self, state_dict, strict=False):
        sd = {(k if k.startswith("perceptual_loss.") else "perceptual_loss." + k): torch.as_tensor(v)
              for k, v in state_dict.items() if "scaling_layer" not in k}
        out = super().load_state_dict(sd, strict=strict)
        self._loaded, self._engine = True, None
        return out

    def _bind(self, device):
        if not self._loaded:
            raise RuntimeError("VQLPIPS has no weights: the reference downloads them (lpips.py:12-48); "
                               "call load_state_dict with the VGG-16 / lin tensors first")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("faceoff_amd.VQLPIPS computes only on an MI355X (cuda/HIP device): there is no CPU fallback")
        if self._engine is None or self._engine.device != device:
            sd = {k[len("perceptual_loss."):]: v for k, v in self.state_dict().items()}
            self._engine = LPIPSEngine(sd, device, dtype=self.compute_dtype)
        return self._engine

    def forward(self, targets, reconstructions):
        return _LPIPSFunction.apply(self, targets.contiguous(), reconstructions.contiguous())

    def per_image(self, inp, target):
        """LPIPS.forward of the reference (lpips.py:80-93): [N,1,1,1], differentiable in `inp`."""
        return _LPIPSPerImageFunction.apply(self, inp.contiguous(), target.contiguous())

    # trainer fast path: loss + gradient accumulated straight into the NHWC decoder-output gradient
    def loss_and_grad(self, gt_nchw, dec_nhwc, g_dec, weight=1.0, taps0=None):
        return self._bind(dec_nhwc.device).loss_and_grad(gt_nchw, dec_nhwc, g_dec, weight, taps0=taps0)

    def target_taps(self, gt_nchw):
        return self._bind(gt_nchw.device).target_taps(gt_nchw)


def _register(root, dotted, tensor):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor))
