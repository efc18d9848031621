"""CPU oracle for the FaceOff hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module.  The product path (faceoff_amd/) never does and fails loudly without its HIP library.

What it is: a functional, CPU-only restatement of the reference's algorithm for the path
BASELINE.json names -- models/vqvae_conv3d_latent.py (Quantize :33-83, ResBlock :86-101,
Encoder :103-131, Decoder :134-166, Conv3dLatentPostnet :169-190, VQVAE :192-295),
models/lpips.py :80-161, loss.py :27-33 and train_faceoff_perceptual.py:32-47,93-107.
The reference is pure PyTorch, so the floating-point primitives (conv / conv_transpose /
matmul / reductions) are the same torch-CPU fp32 primitives the reference dispatches to;
what is restated is everything above them (module wiring, the VQ arithmetic, EMA update,
loss composition, the [B,T] clip generalisation of SURVEY.md section 8 a0).  Parameters are a
plain dict keyed by the reference `state_dict` names (NCHW / OIHW shapes).

Parity pinning: the reference ships no tests or golden vectors, so this
oracle is pinned by outputs of the reference itself, generated in the build container by
tests/golden/make_golden.py (imports /root/reference) and committed under tests/golden/;
tests/test_oracle_golden.py checks the oracle against them, and
tests/test_oracle_vs_reference.py re-checks it live against the imported reference wherever
/root/reference exists (the build container; skipped on the GPU box).
LPIPS: pretrained VGG/lin weights are not obtainable offline => LPIPS parity is on topology
and arithmetic with seeded random weights ("pretrained-weight parity unpinned").
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LATENT_LOSS_WEIGHT = 1.0      # config.py:5
PERCEPTUAL_LOSS_WEIGHT = 1.0  # config.py:6
EMA_DECAY = 0.99              # Quantize.__init__ default (vqvae_conv3d_latent.py:34); VQVAE never forwards `decay` (:209,214)
EMA_EPS = 1e-5


def to_torch_state(sd, requires_grad=True, dtype=torch.float32):
    """dtype: torch.float64 evaluates the SAME restatement in double precision (every function here takes its dtype from its arguments) -- the
    yardstick the timed-size parity tests measure the fp32 reference arithmetic itself against (tests/test_fp64_anchor_gpu.py)."""
    out = OrderedDict()
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v)).clone().to(dtype) if not torch.is_tensor(v) else v.detach().clone().to(dtype).cpu()
        is_buf = k.endswith((".embed", ".cluster_size", ".embed_avg"))
        if requires_grad and not is_buf:
            t.requires_grad_(True)
        out[k] = t
    return out


# --------------------------------------------------------------------------- Quantize
def quantize_forward(x, embed, cluster_size, embed_avg, training, all_reduce=None,
                     decay=EMA_DECAY, eps=EMA_EPS, force_ind=None):
    """Quantize.forward (vqvae_conv3d_latent.py:47-80).  x[..., dim] channels-last.

    Returns (quantize_ste, diff, embed_ind, new_buffers or None).  Buffers are NOT mutated;
    the post-EMA values are returned so callers decide (the reference mutates in place, :66-75).
    `quantize` is gathered from the PRE-update codebook (:57 precedes :59-75).
    """
    dim, n_embed = embed.shape
    flatten = x.reshape(-1, dim)
    dist = (flatten.pow(2).sum(1, keepdim=True) - 2 * flatten @ embed
            + embed.pow(2).sum(0, keepdim=True))                     # :49-53
    _, embed_ind = (-dist).max(1)                                    # :54
    if force_ind is not None:        # teacher-forced codes (parity tests of the bf16-operand engine: both sides use the SAME codes,
        embed_ind = force_ind.reshape(-1).to(torch.int64)   # so a near-tie cannot open an O(1) gap; everything below is :55-78 unchanged)
    embed_onehot = F.one_hot(embed_ind, n_embed).type(flatten.dtype)  # :55
    embed_ind = embed_ind.view(*x.shape[:-1])
    quantize = F.embedding(embed_ind, embed.transpose(0, 1))         # :57,82-83
    new = None
    if training:                                                      # :59
        onehot_sum = embed_onehot.sum(0)
        embed_sum = flatten.detach().transpose(0, 1) @ embed_onehot
        if all_reduce is not None:                                    # :63-64
            onehot_sum = all_reduce(onehot_sum)
            embed_sum = all_reduce(embed_sum)
        cs = cluster_size * decay + onehot_sum * (1 - decay)         # :66-68
        ea = embed_avg * decay + embed_sum * (1 - decay)             # :69
        n = cs.sum()
        csn = (cs + eps) / (n + n_embed * eps) * n                   # :70-73
        new = {"embed": ea / csn.unsqueeze(0), "cluster_size": cs, "embed_avg": ea}   # :74-75
    diff = (quantize.detach() - x).pow(2).mean()                     # :77
    quantize = x + (quantize - x).detach()                           # :78
    return quantize, diff, embed_ind, new


def vq_margin(x, embed):
    """Top-2 distance margin per vector (gate index mismatches on it)."""
    flatten = x.reshape(-1, embed.shape[0]).double()
    e = embed.double()
    dist = flatten.pow(2).sum(1, keepdim=True) - 2 * flatten @ e + e.pow(2).sum(0, keepdim=True)
    top2 = torch.topk(-dist, 2, dim=1).values
    return (top2[:, 0] - top2[:, 1]).float()


class _RoundBF16(torch.autograd.Function):
    """Storage rounding of the bf16 configuration (BASELINE config 3): the value is rounded to bfloat16 on the way
    forward and its gradient is rounded to bfloat16 on the way back (what a bf16 tensor and its bf16 .grad hold);
    all arithmetic around it stays fp32 (= bf16 operands, fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)          # (.to(x.dtype): fp32 as ever; the fp64 evaluation keeps its dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def _identity(x):
    return x


class _RoundGradBF16(torch.autograd.Function):
    """A tensor that stays fp32 on the way forward (the quantiser's input, the decoder output) but whose GRADIENT is stored as
    bfloat16 (it is an operand of the next bf16 data- / filter-gradient GEMM)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def _round_weight(w):
    """bf16 operand copy of an fp32 master filter: value rounded (exactly: the difference is representable), gradient passed
    through unrounded (filter gradients are accumulated and kept in fp32)."""
    return w + (w.bfloat16().to(w.dtype) - w).detach()


class _BF16Sim:
    """Rounding points of the bf16-operand VQ-VAE step (BASELINE config 3 as SURVEY.md section 8(d) defines it: bf16 MFMA
    operands, fp32 accumulation, fp32 master weights, fp32 VQ), shared with faceoff_amd.engine (dtype="bf16"):
      act(x)   every activation a conv kernel stores -- after bias, residual add and ReLU, all done in fp32 on the accumulator --
               is rounded to bf16 ONCE, and so is its gradient (after the ReLU mask and the fan-in add, done in fp32);
      grad(x)  the quantisers' inputs and the decoder output stay fp32 (VQ distances / arg-min / commitment loss and the image
               losses run in fp32 on unrounded values), their gradients are stored as bf16;
      w(w)     filters are rounded to bf16 from the fp32 master copy each step; their gradients stay fp32;
      biases, codebooks, EMA buffers, loss scalars and the optimiser are fp32."""
    act = staticmethod(_RoundBF16.apply)
    grad = staticmethod(_RoundGradBF16.apply)
    w = staticmethod(_round_weight)


class _NoSim:
    act = grad = w = staticmethod(_identity)


class ForcedReLU:
    """Teacher-forced ReLU branches (parity tests at the timed sizes only; the default everywhere is plain F.relu = the reference's
    arithmetic): `relu(x, name)` takes the branch `masks[name]` says (x where True, 0 where False) instead of testing x > 0, and records
    every unit where that differs from its own x > 0 with |x| relative to the tensor's scale -- the near-tie evidence.  Why: ReLU's
    derivative jumps at zero, a pre-activation within fp32 rounding of zero (there are hundreds of millions of units in a 160-frame step:
    a few dozen always are) lets two fp32 implementations take different branches, and a filter or bias gradient -- a sum of ~10^5..10^6
    terms of random sign -- moves by that unit's whole contribution.  With the branches of the implementation under test forced, and each
    difference asserted to be such a near-tie, what is compared is the arithmetic.  The counterpart of `quantize_forward(force_ind=...)`
    and `disc_oracle` `force_masks`.  Site names: see `relu_sites`."""

    def __init__(self, masks):
        self.masks, self.diffs = masks, []

    def __call__(self, x, name):
        m = self.masks.get(name)
        if m is None:
            return F.relu(x)
        xd = x.detach()
        d = m != (xd > 0)
        if d.any():
            self.diffs.append((name, int(d.sum()), float(xd[d].abs().max() / xd.abs().max())))
        return torch.where(m, x, torch.zeros_like(x))


def _plain_relu(x, name):
    return F.relu(x)


# --------------------------------------------------------------------------- conv stacks
def _res_block(x, p, prefix, r=_NoSim, out_relu=False, relu=_plain_relu):
    """ResBlock.forward (:97-101): ReLU -> Conv3x3 -> ReLU -> Conv1x1, out += input.  (out_relu: the Encoder's / Decoder's
    trailing ReLU (:126,145), which the engine applies before the block's output is stored -- the same values either way.)
    relu(x, site): F.relu by default; ForcedReLU in the timed-size parity tests (sites `<prefix>.in`, `.hid`, `.out`)."""
    h = relu(x, prefix + ".in")
    h = F.conv2d(h, r.w(p[prefix + ".conv.1.weight"]), p[prefix + ".conv.1.bias"], padding=1)
    h = r.act(relu(h, prefix + ".hid"))
    h = F.conv2d(h, r.w(p[prefix + ".conv.3.weight"]), p[prefix + ".conv.3.bias"])
    h = h + x
    return r.act(relu(h, prefix + ".out") if out_relu else h)


def encoder(x, p, prefix, stride, n_res_block=2, r=_NoSim, relu=_plain_relu):
    """Encoder (:103-131).  (relu sites: `<prefix>.blocks.0`, `.2` and the ResBlocks')"""
    b = prefix + ".blocks."
    if stride == 4:
        x = r.act(relu(F.conv2d(x, r.w(p[b + "0.weight"]), p[b + "0.bias"], stride=2, padding=1), b + "0"))
        x = r.act(relu(F.conv2d(x, r.w(p[b + "2.weight"]), p[b + "2.bias"], stride=2, padding=1), b + "2"))
        x = r.act(F.conv2d(x, r.w(p[b + "4.weight"]), p[b + "4.bias"], padding=1))
        first = 5
    else:
        x = r.act(relu(F.conv2d(x, r.w(p[b + "0.weight"]), p[b + "0.bias"], stride=2, padding=1), b + "0"))
        x = r.act(F.conv2d(x, r.w(p[b + "2.weight"]), p[b + "2.bias"], padding=1))
        first = 3
    for i in range(n_res_block):
        x = _res_block(x, p, f"{b}{first + i}", r, out_relu=(i == n_res_block - 1), relu=relu)
    return x if n_res_block else F.relu(x)


def decoder(x, p, prefix, stride, n_res_block=2, r=_NoSim, relu=_plain_relu):
    """Decoder (:134-166).  (bf16 policy: the last layer's output is left to the caller -- `dec` itself stays fp32.)
    (relu sites: the ResBlocks' and, stride 4, `<prefix>.blocks.<k>` behind the first transposed convolution)"""
    b = prefix + ".blocks."
    x = r.act(F.conv2d(x, r.w(p[b + "0.weight"]), p[b + "0.bias"], padding=1))
    for i in range(n_res_block):
        x = _res_block(x, p, f"{b}{1 + i}", r, out_relu=(i == n_res_block - 1), relu=relu)
    if not n_res_block:
        x = F.relu(x)
    k = 1 + n_res_block + 1
    x = F.conv_transpose2d(x, r.w(p[f"{b}{k}.weight"]), p[f"{b}{k}.bias"], stride=2, padding=1)
    if stride == 4:
        x = r.act(relu(x, f"{b}{k}"))
        x = F.conv_transpose2d(x, r.w(p[f"{b}{k + 2}.weight"]), p[f"{b}{k + 2}.bias"], stride=2, padding=1)
    return x


def conv3d_postnet(x5, p, prefix, r=_NoSim, relu=_plain_relu):
    """Conv3dLatentPostnet (:169-190) on [B,C,T,H,W].  (relu sites `<prefix>.conv3d.0`, `.1`)"""
    for i in range(3):
        x5 = F.conv3d(x5, r.w(p[f"{prefix}.conv3d.{i}.0.weight"]), p[f"{prefix}.conv3d.{i}.0.bias"], padding=1)
        if i < 2:
            x5 = relu(x5, f"{prefix}.conv3d.{i}")
        x5 = r.act(x5)
    return x5


def vqvae_forward(x, p, training=True, all_reduce=None, T=None, bf16sim=False, force_ids=None, relu=None):
    """VQVAE.forward (:243-259) generalised to clips.

    x: [B,T,6,H,W] (or [N,6,H,W] with T=None => one clip of N frames, the literal reference).
    Returns dict(dec[N,6,H,W], diff[1], id_t, id_b, new_buffers, enc/quant intermediates).
    """
    if x.dim() == 5:
        B, T_, C, H, W = x.shape
        frames = x.reshape(B * T_, C, H, W)
    else:
        frames = x
        T_ = T if T is not None else x.shape[0]
        B = x.shape[0] // T_
    r = _BF16Sim if bf16sim else _NoSim      # (bf16sim: the rounding points of the bf16-operand engine, see _BF16Sim)
    relu = relu or _plain_relu               # (a ForcedReLU in the timed-size parity tests; F.relu = the reference otherwise)
    enc_b = encoder(r.act(frames) if bf16sim else frames, p, "enc_b", 4, r=r, relu=relu)                    # :237-241
    enc_t = encoder(r.act(enc_b), p, "enc_t", 2, r=r, relu=relu)        # (r.act on a stored tensor: a no-op forward; backward it rounds THIS consumer's gradient, which the engine stores before the fan-in add)

    def clips(t):   # :247  [N,C,h,w] -> [B,C,T,h,w]
        n, c, h, w = t.shape
        return t.reshape(B, T_, c, h, w).permute(0, 2, 1, 3, 4)

    def frames_of(t5):  # :251
        b, c, tt, h, w = t5.shape
        return t5.permute(0, 2, 1, 3, 4).reshape(b * tt, c, h, w)

    enc_b_conv = frames_of(conv3d_postnet(clips(enc_b), p, "conv3d_encoded_b", r, relu))   # :250
    enc_t_conv = frames_of(conv3d_postnet(clips(enc_t), p, "conv3d_encoded_t", r, relu))

    # encode_quantized (:261-278)
    qt_in = r.grad(F.conv2d(enc_t_conv, r.w(p["quantize_conv_t.weight"]), p["quantize_conv_t.bias"]).permute(0, 2, 3, 1))
    quant_t, diff_t, id_t, new_t = quantize_forward(
        qt_in, p["quantize_t.embed"], p["quantize_t.cluster_size"], p["quantize_t.embed_avg"],
        training, all_reduce, force_ind=None if force_ids is None else force_ids[0])
    quant_t = r.act(quant_t.permute(0, 3, 1, 2))
    dec_t = r.act(decoder(quant_t, p, "dec_t", 2, r=r, relu=relu))
    cat_b = torch.cat([dec_t, enc_b_conv], 1)
    qb_in = r.grad(F.conv2d(cat_b, r.w(p["quantize_conv_b.weight"]), p["quantize_conv_b.bias"]).permute(0, 2, 3, 1))
    quant_b, diff_b, id_b, new_b = quantize_forward(
        qb_in, p["quantize_b.embed"], p["quantize_b.cluster_size"], p["quantize_b.embed_avg"],
        training, all_reduce, force_ind=None if force_ids is None else force_ids[1])
    quant_b = r.act(quant_b.permute(0, 3, 1, 2))
    diff = diff_t.unsqueeze(0) + diff_b.unsqueeze(0)

    # decode (:280-285)
    upsample_t = r.act(F.conv_transpose2d(r.act(quant_t), r.w(p["upsample_t.weight"]), p["upsample_t.bias"], stride=2, padding=1))
    dec = r.grad(decoder(torch.cat([upsample_t, quant_b], 1), p, "dec", 4, r=r, relu=relu))
    new_buffers = None
    if training:
        new_buffers = {f"quantize_t.{k}": v for k, v in new_t.items()}
        new_buffers.update({f"quantize_b.{k}": v for k, v in new_b.items()})
    return dict(dec=dec, diff=diff, id_t=id_t, id_b=id_b, new_buffers=new_buffers,
                qt_in=qt_in, qb_in=qb_in, enc_b=enc_b, enc_t=enc_t,
                enc_b_conv=enc_b_conv, enc_t_conv=enc_t_conv, quant_t=quant_t, quant_b=quant_b)


# --------------------------------------------------------------------------- LPIPS
_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
_VGG_SLICE_OF = {}
for _s, (_a, _b) in enumerate([(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)], start=1):
    for _i in range(_a, _b):
        _VGG_SLICE_OF[_i] = _s
LPIPS_SHIFT = (-.030, -.088, -.188)   # lpips.py:99
LPIPS_SCALE = (.458, .448, .450)      # lpips.py:100


def vgg16_taps(x, lp, bf16sim=False):
    """vgg16.forward (lpips.py:139-152): relu1_2, relu2_2, relu3_3, relu4_3, relu5_3.
    bf16sim: every stored activation (and its gradient) and every filter is rounded to bfloat16."""
    taps, idx = [], 0
    tap_after = {3, 8, 15, 22, 29}   # last ReLU index of each slice (:125-134)
    rnd = _RoundBF16.apply if bf16sim else _identity
    x = rnd(x)
    for v in _VGG_CFG:
        if v == "M":
            x = rnd(F.max_pool2d(x, 2, 2))
            idx += 1
            continue
        s = _VGG_SLICE_OF[idx]
        w = lp[f"net.slice{s}.{idx}.weight"]
        if bf16sim:
            w = w.bfloat16().to(w.dtype)
        x = rnd(F.relu(F.conv2d(x, w, lp[f"net.slice{s}.{idx}.bias"], padding=1)))
        idx += 2
        if idx - 1 in tap_after:
            taps.append(x)
    return taps


def lpips_forward(inp, target, lp, per_tap=False, bf16sim=False):
    """LPIPS.forward (lpips.py:80-93) -> [N,1,1,1]; Dropout is identity in eval (loss.py:30)."""
    shift = torch.tensor(LPIPS_SHIFT, dtype=inp.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE, dtype=inp.dtype).view(1, 3, 1, 1)
    f0 = vgg16_taps((inp - shift) / scale, lp, bf16sim)
    f1 = vgg16_taps((target - shift) / scale, lp, bf16sim)
    res = []
    for kk in range(5):
        n0 = f0[kk] / (torch.sqrt(torch.sum(f0[kk] ** 2, dim=1, keepdim=True)) + 1e-10)   # :155-157
        n1 = f1[kk] / (torch.sqrt(torch.sum(f1[kk] ** 2, dim=1, keepdim=True)) + 1e-10)
        d = (n0 - n1) ** 2
        res.append(F.conv2d(d, lp[f"lin{kk}.model.1.weight"]).mean([2, 3], keepdim=True))   # :89,160-161
    val = res[0]
    for l in range(1, 5):
        val = val + res[l]
    return (val, res) if per_tap else val


# --------------------------------------------------------------------------- the step
def run_step(x, ground_truth, p, lpips_state=None, training=True, all_reduce=None, lpips_bf16sim=False, bf16sim=False, force_ids=None,
             weights=(1.0, LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT), relu=None):
    """run_step + loss composition (train_faceoff_perceptual.py:32-47,97-98).

    x[B,T,6,H,W], ground_truth[B,T,3,H,W].  Returns dict with recon/latent/perceptual/loss
    and the forward dict.  perceptual is 0 when lpips_state is None (BASELINE config 2).
    """
    fw = vqvae_forward(x, p, training=training, all_reduce=all_reduce, bf16sim=bf16sim, force_ids=force_ids, relu=relu)
    gt = ground_truth.reshape(-1, *ground_truth.shape[-3:])
    out = fw["dec"][:, :3]                                   # :37
    recon = F.mse_loss(out, gt)                              # :21,39
    latent = fw["diff"].mean()                               # :40
    if lpips_state is not None:
        perceptual = lpips_forward(gt.contiguous(), out.contiguous(), lpips_state, bf16sim=lpips_bf16sim).mean()   # loss.py:33
    else:
        perceptual = torch.zeros((), dtype=recon.dtype)
    loss = weights[0] * recon + weights[1] * latent + weights[2] * perceptual   # :98 (weights: (1, 1, 1) in the reference; tests isolate a term)
    return dict(recon=recon, latent=latent, perceptual=perceptual, loss=loss, fw=fw)


def adam_step(p, grads, state, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (train_faceoff_perceptual.py:190), restated per tensor."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for k, g in grads.items():
        m = state.setdefault("m." + k, torch.zeros_like(g))
        v = state.setdefault("v." + k, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / np.sqrt(1 - b2 ** t)).add_(eps)
        with torch.no_grad():
            p[k].addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def train_step(x, ground_truth, p, lpips_state=None, adam_state=None, lr=3e-4, lpips_bf16sim=False, bf16sim=False, force_ids=None,
               weights=(1.0, LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT)):
    """One iteration of train() (:93-107): zero_grad, run_step, backward, (Adam), EMA buffers."""
    params = {k: v for k, v in p.items() if v.requires_grad}
    for v in params.values():
        v.grad = None
    r = run_step(x, ground_truth, p, lpips_state, training=True, lpips_bf16sim=lpips_bf16sim, bf16sim=bf16sim, force_ids=force_ids, weights=weights)
    r["loss"].backward()
    grads = {k: v.grad.detach().clone() for k, v in params.items()}
    with torch.no_grad():
        for k, v in r["fw"]["new_buffers"].items():
            p[k].copy_(v)
    if adam_state is not None:
        adam_step(p, grads, adam_state, lr=lr)
    r["grads"] = grads
    return r


# --------------------------------------------------------------------------- per-op oracles (NHWC views)
def nhwc_conv2d(x_nhwc, w_oihw, bias, stride, padding):
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_oihw, bias, stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1).contiguous()


def nhwc_conv_transpose2d(x_nhwc, w_iohw, bias, stride=2, padding=1):
    y = F.conv_transpose2d(x_nhwc.permute(0, 3, 1, 2), w_iohw, bias, stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1).contiguous()


def ndhwc_conv3d(x_bthwc, w_oidhw, bias, padding=1):
    y = F.conv3d(x_bthwc.permute(0, 4, 1, 2, 3), w_oidhw, bias, padding=padding)
    return y.permute(0, 2, 3, 4, 1).contiguous()
