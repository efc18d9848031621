"""Drop-in mirrors of the reference's `ModelD_3d` (TemporalAlignment/models/mocoganhd_video_disc.py:8-30) and `ModelD_img`
(mocoganhd_content_disc.py:8-24) on the gfx950 kernels: same constructor arguments, `forward(x)` -> list (per scale) of
lists (per layer) of feature maps in NCDHW / NCHW, the same 38 state_dict keys (`netD.scale{i}_layer{j}.0.weight` ...,
InstanceNorm running statistics), and `.optim` = Adam(lr, betas=(0.5, 0.999)) over the parameters, so the reference's
`modelD.optim.zero_grad(); loss.backward(); modelD.optim.step()` (train_vqvae_mocoganhd_disc.py:404-432) runs unchanged.

Autograd: the whole multiscale network is one node; gradients flow through the LAST feature map of each scale (the patch
logits -- the only ones the trainer's Relativistic_Average_LSGAN reads, mocoganhd_losses.py:117-118); the intermediate feature
maps are returned detached.  Only norm_D_3d='instance' (what the file's default and every reachable call path use)."""
from __future__ import annotations

import torch
from torch import nn

from ..disc import DiscEngine
from ..synth import disc_param_specs


def _register(root, dotted, tensor, buffer=False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class _DiscFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._bind(x.device)
        eng.mark_params_dirty()       # the module's parameters are stepped by torch.optim through .data aliases of the arena: always re-pack here
        five = x if x.dim() == 5 else x.unsqueeze(2)                  # [N,C,D,H,W]
        N, Cc, D, H, W = five.shape
        xc = torch.zeros((N, D, H, W, 32), device=x.device)
        xc[..., :Cc] = five.permute(0, 2, 3, 4, 1)
        S = eng.forward(xc.contiguous(), training=model.training)
        ctx.model, ctx.S, ctx.x_shape, ctx.needs_x = model, S, x.shape, x.requires_grad
        outs = []
        for sc in S["scales"]:
            for j, f in enumerate(sc["feat"]):
                co = f.shape[-1] if j < 4 else 1
                t = f[..., :co].permute(0, 4, 1, 2, 3)
                outs.append((t if x.dim() == 5 else t.squeeze(2)).contiguous())
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i % 5 != 4])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        model, S = ctx.model, ctx.S
        eng = model._engine
        g_logits = []
        for i, lg in enumerate(S["logits"]):
            g = torch.zeros_like(lg)
            go = gouts[5 * i + 4]
            if go is not None:
                g5 = go if go.dim() == 5 else go.unsqueeze(2)
                g[..., 0] = g5[:, 0]
            g_logits.append(g)
        gx = eng.backward(S, g_logits, param_grads=True, input_grad=ctx.needs_x)
        grads = tuple(eng.grads[k].clone() for k in model._param_keys)
        gin = None
        if ctx.needs_x:
            Cc = ctx.x_shape[1]
            gin = gx[..., :Cc].permute(0, 4, 1, 2, 3)
            gin = gin.reshape(ctx.x_shape).contiguous() if len(ctx.x_shape) == 5 else gin.squeeze(2).contiguous()
        return (None, gin) + grads


class _ModelD(nn.Module):
    def __init__(self, dims, nc, num_D, lr, n_frames):
        super().__init__()
        self.dims, self.nc, self.num_D, self.n_frames = dims, nc, num_D, n_frames
        self._param_keys = []
        fan_in = 1
        for key, shape in disc_param_specs(dims, nc, num_D):
            if key.endswith(".weight"):
                fan_in = int(torch.tensor(shape[1:]).prod())
                _register(self, key, torch.empty(shape).normal_(0.0, 0.02))                       # weights_init (:32-35)
                self._param_keys.append(key)
            elif key.endswith(".bias"):
                b = 1.0 / fan_in ** 0.5
                _register(self, key, torch.empty(shape).uniform_(-b, b))
                self._param_keys.append(key)
            elif key.endswith("running_mean"):
                _register(self, key, torch.zeros(shape), buffer=True)
            elif key.endswith("running_var"):
                _register(self, key, torch.ones(shape), buffer=True)
            else:
                _register(self, key, torch.zeros(shape, dtype=torch.int64), buffer=True)
        self._engine = None
        self.optim = torch.optim.Adam(self.parameters(), lr=lr, betas=(0.5, 0.999))               # :24-26

    def _bind(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("faceoff_amd discriminators compute only on an MI355X (cuda/HIP device): there is no CPU fallback")
        named, bufs = dict(self.named_parameters()), dict(self.named_buffers())
        eng = self._engine
        if eng is not None and eng.device == device and all(named[k].data_ptr() == eng.params[k].data_ptr() for k in self._param_keys):
            return eng
        sd = {k: v.detach() for k, v in list(named.items()) + list(bufs.items())}
        eng = DiscEngine(sd, device, dims=self.dims, nc=self.nc, num_D=self.num_D, n_frames=self.n_frames)
        for k in self._param_keys:
            named[k].data = eng.params[k]
        for k, b in bufs.items():
            b.data = eng.buffers[k]
        self._engine = eng
        return eng

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        dev = next(self.parameters()).device
        if dev.type == "cuda":
            self._bind(dev)
        return out

    def forward(self, x):
        flat = _DiscFunction.apply(self, x.float(), *[p for _, p in self.named_parameters()])
        return [list(flat[5 * i:5 * i + 5]) for i in range(self.num_D)]


class ModelD_3d(_ModelD):
    def __init__(self, nc, norm_D_3d, num_D, lr, cross_domain, n_frames_G):
        if norm_D_3d != "instance":
            raise NotImplementedError("only norm_D_3d='instance' is built (the reference's default)")
        if not cross_domain:                                                                      # :11-16
            nc, n_frames_G = nc * 2, n_frames_G - 1
        super().__init__(3, nc, num_D, lr, n_frames_G)


class ModelD_img(_ModelD):
    def __init__(self, nc, norm_D_3d, num_D, lr):
        if norm_D_3d != "instance":
            raise NotImplementedError("only norm_D_3d='instance' is built (the reference's default)")
        super().__init__(2, nc * 2, num_D, lr, 16)
