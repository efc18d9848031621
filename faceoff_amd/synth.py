"""Synthetic weights and inputs for the FaceOff hot path (numpy-seeded, torch-RNG independent).

The reference has no test data, so every parity case is built from
`numpy.random.default_rng(seed)` streams; the same arrays are produced in the golden
generator (tests/golden/make_golden.py, run beside the reference), in the tests and in
bench.py, on any machine.

Shapes/keys follow the reference `state_dict` (models/vqvae_conv3d_latent.py:193-231).
Init bounds restate torch's defaults for nn.Conv*/ConvTranspose* (kaiming_uniform(a=sqrt(5))
=> U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias), codebooks follow
Quantize.__init__ (vqvae_conv3d_latent.py:42-45) with an optional scale (SURVEY.md section 7:
the default randn codebook is degenerate against an untrained encoder).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# (key prefix, kind, shape) in reference state_dict order.  kind: conv | convT | conv3d
def vqvae_param_specs(in_channel=6, channel=128, n_res_block=2, n_res_channel=32,
                      embed_dim=64, n_embed=512):
    s = []

    def conv(name, co, ci, k):
        s.append((name, "conv", (co, ci, k, k)))

    def convT(name, ci, co, k):
        s.append((name, "convT", (ci, co, k, k)))

    def res(prefix):
        conv(prefix + ".conv.1", n_res_channel, channel, 3)
        conv(prefix + ".conv.3", channel, n_res_channel, 1)

    # enc_b: Encoder stride 4 (vqvae_conv3d_latent.py:107-114)
    conv("enc_b.blocks.0", channel // 2, in_channel, 4)
    conv("enc_b.blocks.2", channel, channel // 2, 4)
    conv("enc_b.blocks.4", channel, channel, 3)
    for i in range(n_res_block):
        res(f"enc_b.blocks.{5 + i}")
    # enc_t: Encoder stride 2 (:116-121)
    conv("enc_t.blocks.0", channel // 2, channel, 4)
    conv("enc_t.blocks.2", channel, channel // 2, 3)
    for i in range(n_res_block):
        res(f"enc_t.blocks.{3 + i}")
    conv("quantize_conv_t", embed_dim, channel, 1)
    s.append(("quantize_t", "vq", (embed_dim, n_embed)))
    # dec_t: Decoder stride 2 (:140-161)
    conv("dec_t.blocks.0", channel, embed_dim, 3)
    for i in range(n_res_block):
        res(f"dec_t.blocks.{1 + i}")
    convT(f"dec_t.blocks.{2 + n_res_block}", channel, embed_dim, 4)
    conv("quantize_conv_b", embed_dim, embed_dim + channel, 1)
    s.append(("quantize_b", "vq", (embed_dim, n_embed)))
    convT("upsample_t", embed_dim, embed_dim, 4)
    # dec: Decoder stride 4
    conv("dec.blocks.0", channel, embed_dim + embed_dim, 3)
    for i in range(n_res_block):
        res(f"dec.blocks.{1 + i}")
    convT(f"dec.blocks.{2 + n_res_block}", channel, channel // 2, 4)
    convT(f"dec.blocks.{4 + n_res_block}", channel // 2, in_channel, 4)
    for lvl in ("b", "t"):
        for i in range(3):
            s.append((f"conv3d_encoded_{lvl}.conv3d.{i}.0", "conv3d", (128, 128, 3, 3, 3)))
    return s


def make_state_dict(seed=0, codebook_scale=1.0, in_channel=6, gain=1.0, codebook_center=None, **kw):
    """Reference-keyed state dict of float32 numpy arrays (70 params + 6 buffers).
    codebook_center: optional {"quantize_t": vec[64], "quantize_b": vec[64]} added to every code of that level (an
    untrained encoder's latents sit far from the origin; a codebook centred on them uses hundreds of codes instead of
    a few dozen -- the golden fixtures store the centres they were generated with)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, kind, shape in vqvae_param_specs(in_channel=in_channel, **kw):
        if kind == "vq":
            e = (rng.standard_normal(shape) * codebook_scale).astype(np.float32)
            if codebook_center is not None and codebook_center.get(name) is not None:
                e = (e + np.asarray(codebook_center[name], np.float32).reshape(-1, 1)).astype(np.float32)
            sd[name + ".embed"] = e
            sd[name + ".cluster_size"] = np.zeros(shape[1], np.float32)
            sd[name + ".embed_avg"] = e.copy()
            continue
        # torch _calculate_fan_in_and_fan_out: fan_in = size(1) * receptive field
        fan_in = shape[1] * int(np.prod(shape[2:]))
        bound = gain / math.sqrt(fan_in)
        sd[name + ".weight"] = rng.uniform(-bound, bound, shape).astype(np.float32)
        nb = shape[1] if kind == "convT" else shape[0]
        sd[name + ".bias"] = rng.uniform(-bound, bound, (nb,)).astype(np.float32)
    return sd


def golden_state(g):
    """The state dict a golden fixture (tests/golden/*.npz) was generated with: seeds, scales and -- where the
    fixture holds them -- the codebook centres."""
    center = None
    if "codebook_center_t" in g.files:
        center = {"quantize_t": g["codebook_center_t"], "quantize_b": g["codebook_center_b"]}
    return make_state_dict(int(g["seed_w"]), codebook_scale=float(g["codebook_scale"]), gain=float(g["gain"]),
                           codebook_center=center)


def make_batch(seed, B, T, H, W):
    """Loader 5-tuple restated (utils.py:29-38): returns img[B,T,6,H,W], ground_truth[B,T,3,H,W]
    in U(-1,1) (TemporalAlignment/dataset.py:240-247 normalises to [-1,1])."""
    rng = np.random.default_rng(seed)
    source = rng.uniform(-1, 1, (B, T, 3, H, W)).astype(np.float32)
    background = rng.uniform(-1, 1, (B, T, 3, H, W)).astype(np.float32)
    source_images = rng.uniform(-1, 1, (B, T, 3, H, W)).astype(np.float32)
    img = np.concatenate([source, background], axis=2)
    return img, source_images


def make_vgg_lpips_state(seed=0):
    """Seeded stand-in for torchvision VGG-16 features + LPIPS lin weights (not obtainable
    offline: SURVEY.md section 8c).  Keys follow models/lpips.py (net.sliceK.<idx>.weight, linK.model.1.weight)."""
    rng = np.random.default_rng(seed)
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
    slices = [(1, 0, 4), (2, 4, 9), (3, 9, 16), (4, 16, 23), (5, 23, 30)]
    sd = OrderedDict()
    idx, cin = 0, 3
    for v in cfg:
        if v == "M":
            idx += 1
            continue
        sl = next(s for s, a, b in slices if a <= idx < b)
        std = math.sqrt(2.0 / (cin * 9))
        sd[f"net.slice{sl}.{idx}.weight"] = (rng.standard_normal((v, cin, 3, 3)) * std).astype(np.float32)
        sd[f"net.slice{sl}.{idx}.bias"] = (rng.standard_normal((v,)) * 0.05).astype(np.float32)
        cin = v
        idx += 2
    for k, c in enumerate([64, 128, 256, 512, 512]):
        sd[f"lin{k}.model.1.weight"] = rng.uniform(0, 1.0 / c, (1, c, 1, 1)).astype(np.float32)
    return sd


DISC_CHANNELS = (64, 128, 256, 512, 1)      # NLayerDiscriminator(ndf=64, n_layers=3): mocoganhd_video_disc.py:133-158


def disc_param_specs(dims=3, nc=6, num_D=2):
    """(key, shape) of every parameter / buffer of ModelD_3d (dims=3) or ModelD_img (dims=2) in reference state_dict order
    (TemporalAlignment/models/mocoganhd_video_disc.py:56-76: scale{i}_layer{j}.0 = conv, .1 = InstanceNorm with running stats)."""
    out = []
    for i in range(num_D):
        cin = nc
        for j, co in enumerate(DISC_CHANNELS):
            key = f"netD.scale{i}_layer{j}"
            out.append((key + ".0.weight", (co, cin) + (4,) * dims))
            out.append((key + ".0.bias", (co,)))
            if 1 <= j <= 3:
                out.append((key + ".1.running_mean", (co,)))
                out.append((key + ".1.running_var", (co,)))
                out.append((key + ".1.num_batches_tracked", ()))
            cin = co
    return out


def make_disc_state(seed=0, dims=3, nc=6, num_D=2, weight_std=0.02):
    """Seeded discriminator state: conv weights N(0, 0.02) (weights_init, mocoganhd_video_disc.py:32-38), biases torch's
    default U(-1/sqrt(fan_in), 1/sqrt(fan_in)), running_mean 0, running_var 1, num_batches_tracked 0."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    fan_in = None
    for key, shape in disc_param_specs(dims, nc, num_D):
        if key.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            sd[key] = (rng.standard_normal(shape) * weight_std).astype(np.float32)
        elif key.endswith(".bias"):
            b = 1.0 / math.sqrt(fan_in)
            sd[key] = rng.uniform(-b, b, shape).astype(np.float32)
        elif key.endswith("running_mean"):
            sd[key] = np.zeros(shape, np.float32)
        elif key.endswith("running_var"):
            sd[key] = np.ones(shape, np.float32)
        else:
            sd[key] = np.zeros(shape, np.int64)
    return sd
