"""The kernels of the bf16-operand VQ-VAE step (BASELINE config 3 as SURVEY.md 8(d) defines it), each through the C ABI against torch-CPU
fp32 convolutions of the SAME bf16-rounded operands (products of bf16 numbers are exact in fp32, so what differs is the fp32 summation
order and the one final rounding of a stored bf16 result):
  * bf16 outputs: within one bf16 ulp of the fp32 reference (2^-8 relative) + 1e-3 of the tensor's scale for cancelling sums;
  * fp32 outputs (quantiser inputs, decoder output, filter / bias gradients): 2e-5 of the tensor's scale.
Geometries: every conv form of models/vqvae_conv3d_latent.py:92-190 -- Conv3d k3 (T = 1, 2, 3, 5), k3 s1, k4 s2, 1x1 (32, 128 and 192 input
channels), the 8-channel image layer, transposed k4 s2 as four phases and as the one-launch cell form -- forward, data gradient with ReLU
mask / fan-in add, and filter gradient in both its forms (row runs; gather for widths that are not multiples of 32)."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(t):
    return t.bfloat16().float()


def nhwc(t):            # [N,C,H,W] fp32 (bf16-representable) -> bf16 channels-last on the GPU
    return t.permute(0, 2, 3, 1).contiguous().to(BF).cuda()


def back(t):            # channels-last GPU tensor -> [N,C,H,W] fp32 on the CPU
    return t.float().cpu().permute(0, 3, 1, 2)


def close_bf16(got, ref, what=""):
    got, ref = got.float(), ref.float()
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} off, worst {((got - ref).abs() / tol).max().item():.2f} x tol"


def close_f32(got, ref, tol=2e-5, what=""):
    err = (got.float() - ref.float()).abs().max().item() / (ref.abs().max().item() + 1e-30)
    assert err <= tol, f"{what}: rel err {err:.2e}"


def gen(seed):
    return torch.Generator().manual_seed(seed)


def packed_bf16(w_packed_f32):
    from faceoff_amd import ops
    return ops.to_bf16(w_packed_f32)


# ------------------------------------------------------------------------------------------------ forward / data gradient
@pytest.mark.parametrize("T", [1, 2, 3, 5])
@pytest.mark.parametrize("big", [False, True])
def test_conv3d_forward_and_data_gradient(T, big, monkeypatch):
    """nn.Conv3d(128, 128, 3, padding=1) (:181) on clips of T frames: forward with bias + ReLU, data gradient with ReLU mask + fan-in add,
    on the 128-row tiles and (FACEOFF_BF16_BIG_TILES) on the 256-row ping-pong kernel that the C2 shapes take (both skip the depth taps
    that only see clip padding)."""
    from faceoff_amd import ops
    if big:
        monkeypatch.setenv("FACEOFF_BF16_BIG_TILES", "1")
    g = gen(10 + T)
    B, H, W, Cc = 2, 16, 16, 128
    N = B * T
    x = rb(torch.randn((N, Cc, H, W), generator=g))
    w = rb(torch.randn((Cc, Cc, 3, 3, 3), generator=g) / np.sqrt(27 * Cc))
    b = torch.randn(Cc, generator=g)
    x5 = x.reshape(B, T, Cc, H, W).permute(0, 2, 1, 3, 4)
    ref = F.relu(F.conv3d(x5, w, b, padding=1)).permute(0, 2, 1, 3, 4).reshape(N, Cc, H, W)
    wp = packed_bf16(ops.pack_conv(w.reshape(Cc, Cc, 27).cuda()))
    out = torch.empty((N, H, W, Cc), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(x), wp, b.cuda(), out, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=Cc, cout=Cc, flags=ops.FO_OUT_RELU)
    close_bf16(back(out), ref, "conv3d forward")
    # data gradient: conv of the output gradient with the flipped, transposed filter; * (mask > 0) + add
    gy = rb(torch.randn((N, Cc, H, W), generator=g))
    mask = rb(torch.randn((N, Cc, H, W), generator=g))
    add = rb(torch.randn((N, Cc, H, W), generator=g))
    gy5 = gy.reshape(B, T, Cc, H, W).permute(0, 2, 1, 3, 4)
    gref = F.conv_transpose3d(gy5, w, padding=1).permute(0, 2, 1, 3, 4).reshape(N, Cc, H, W) * (mask > 0) + add
    wpd = packed_bf16(ops.pack_conv_dgrad(w.reshape(Cc, Cc, 27).cuda()))
    gin = torch.empty((N, H, W, Cc), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(gy), wpd, None, gin, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=Cc, cout=Cc, mask=nhwc(mask), add=nhwc(add))
    close_bf16(back(gin), gref, "conv3d data gradient")


@pytest.mark.parametrize("cin,cout,H", [(64, 128, 32), (128, 64, 16), (8, 64, 64)])
def test_conv_k4s2_forward(cin, cout, H):
    """nn.Conv2d(k=4, s=2, p=1) (:109,111,118): the stems, incl. the 8-channel image layer"""
    from faceoff_amd import ops
    g = gen(cin + cout)
    N, creal = 3, (6 if cin == 8 else cin)
    x = torch.zeros((N, cin, H, H))
    x[:, :creal] = rb(torch.randn((N, creal, H, H), generator=g))
    w = rb(torch.randn((cout, creal, 4, 4), generator=g) / np.sqrt(16 * creal))
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x[:, :creal], w, b, stride=2, padding=1))
    wp = packed_bf16(ops.pack_conv(w.cuda()))
    out = torch.empty((N, H // 2, H // 2, cout), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(x), wp, b.cuda(), out, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=cin, cout=cout, flags=ops.FO_OUT_RELU)
    close_bf16(back(out), ref, f"k4s2 {cin}->{cout}")


def test_transposed_conv_phases_into_a_channel_slice_and_its_data_gradient():
    """nn.ConvTranspose2d(128, 64, 4, stride=2, padding=1) (:157) as four sub-pixel launches writing channels 0..63 of a 192-channel
    buffer (torch.cat([dec_t, enc_b], 1), :271); and its data gradient = a k4 s2 conv of the output gradient slice, with ReLU mask."""
    from faceoff_amd import ops
    g = gen(3)
    N, ci, co, h = 2, 128, 64, 16
    x = rb(torch.randn((N, ci, h, h), generator=g))
    w = rb(torch.randn((ci, co, 4, 4), generator=g) / np.sqrt(4 * ci))
    b = torch.randn(co, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    cat = torch.full((N, 2 * h, 2 * h, 192), 7.0, device="cuda", dtype=BF)
    ops.convT_phases_bf16(nhwc(x), packed_bf16(ops.pack_convT(w.cuda())), b.cuda(), cat[..., 0:64], cin=ci, cout=co)
    close_bf16(back(cat[..., 0:64]), ref, "convT phases")
    assert bool((cat[..., 64:] == 7.0).all())
    gy = rb(torch.randn((N, 192, 2 * h, 2 * h), generator=g))
    mask = rb(torch.randn((N, ci, h, h), generator=g))
    gref = F.conv2d(gy[:, :64], w, stride=2, padding=1) * (mask > 0)
    gin = torch.empty((N, h, h, ci), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(gy)[..., 0:64], packed_bf16(ops.pack_conv(w.cuda())), None, gin, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=co, cout=ci,
                   mask=nhwc(mask))
    close_bf16(back(gin), gref, "convT data gradient")


@pytest.mark.parametrize("big", [0, 1, 2])
@pytest.mark.parametrize("ci,co,h,w", [(128, 64, 16, 16), (64, 64, 5, 7), (64, 128, 6, 10), (128, 40, 4, 4)])
def test_transposed_conv_as_one_cell_form_launch_at_any_width(ci, co, h, w, big, monkeypatch):
    """The k4 s2 p1 transposed convolutions with more than 8 output channels (dec.blocks.4, dec_t.blocks.4, upsample_t :147-151,222) and the data gradients
    of the k4 s2 convolutions (enc_b.blocks.2, enc_t.blocks.0) as ONE launch: a k2 p1 conv over the (H+1) x (W+1) cell grid, 4 x Cpp GEMM columns,
    depth-to-space in the epilogue (fo_pack_convT_k4s2_cells_n + FO_DEPTH2SPACE).  Forward use: bias + ReLU into a channel slice of a wider buffer; gradient
    use: ReLU mask and fan-in add at the output pixel.  Against torch on the same bf16 operands, and against the four sub-pixel phase launches."""
    from faceoff_amd import ops, _lib
    # big: 0 the 128-row kernel (what these small sizes get), 1 the per-tap big-tile kernel at 256 x 256 (the timed sizes), 2 at 256 x 128
    if big:
        monkeypatch.setenv("FACEOFF_BF16_BIG_TILES", "1")
    if big == 2:
        monkeypatch.setenv("FACEOFF_BF16_CELLS_NO_256", "1")
    g = gen(300 + ci + co + h)
    N = 2
    x = rb(torch.randn((N, ci, h, w), generator=g))
    wt = rb(torch.randn((ci, co, 4, 4), generator=g) / np.sqrt(4 * ci))
    lib = _lib.load()
    lib.fo_kernel_notes(1); lib.fo_last_kernel()
    b = torch.randn(co, generator=g)
    wc = packed_bf16(ops.pack_convT_cells(wt.cuda()))
    ld = (co + 7) // 8 * 8 + 64
    ref = F.relu(F.conv_transpose2d(x, wt, b, stride=2, padding=1))
    cat = torch.full((N, 2 * h, 2 * w, ld), 7.0, device="cuda", dtype=BF)
    ops.convT_cells_bf16(nhwc(x), wc, b.cuda(), cat[..., 0:(co + 7) // 8 * 8], cin=ci, cout=co, flags=ops.FO_OUT_RELU)
    torch.cuda.synchronize()
    kern = lib.fo_last_kernel().decode(); lib.fo_kernel_notes(0)
    assert ("conv_bf16_pp16_kernel<256, 256" in kern) if big == 1 else ("conv_bf16_pp16_kernel<256, 128" in kern) if big == 2 else ("pp16" not in kern), kern
    close_bf16(back(cat[..., 0:co]), ref, "convT cells forward")
    assert bool((cat[..., (co + 7) // 8 * 8:] == 7.0).all())
    # gradient use: masked, added to a fan-in gradient
    mask = rb(torch.randn((N, co, 2 * h, 2 * w), generator=g))
    addt = rb(torch.randn((N, co, 2 * h, 2 * w), generator=g))
    gref = F.conv_transpose2d(x, wt, None, stride=2, padding=1) * (mask > 0) + addt
    def pad8(t):          # NHWC with the channel count rounded up to 8
        o = torch.zeros((N, 2 * h, 2 * w, (co + 7) // 8 * 8), device="cuda", dtype=BF)
        o[..., :co] = nhwc(t)
        return o
    gin = torch.empty((N, 2 * h, 2 * w, (co + 7) // 8 * 8), device="cuda", dtype=BF)
    ops.convT_cells_bf16(nhwc(x), wc, None, gin, cin=ci, cout=co, mask=pad8(mask), add=pad8(addt))
    close_bf16(back(gin[..., :co]), gref, "convT cells as a data gradient")
    if co % 32 == 0:      # the phase launches on the same operands: the same products in another order
        gin4 = torch.empty_like(gin)
        ops.convT_phases_bf16(nhwc(x), packed_bf16(ops.pack_convT(wt.cuda())), None, gin4, cin=ci, cout=co, mask=pad8(mask), add=pad8(addt))
        d = (gin.float() - gin4.float()).abs()
        tol = gin4.float().abs() * 2.0 ** -7 + 1e-3 * gin4.float().abs().max()
        assert not bool((d > tol).any()), f"cells vs phases: worst {float((d / tol).max()):.2f} x tol"


def test_resblock_pair_of_launches():
    """ResBlock (:86-101): h = relu(conv3x3(relu(x)) + b1) with the leading ReLU applied as the operand is staged (FO_IN_RELU, 128 -> 32),
    out = relu(conv1x1(h) + b3 + x) (32 input channels: the 256-row kernel's 32-deep K tiles; residual add + trailing ReLU in the epilogue)."""
    from faceoff_amd import ops
    g = gen(4)
    N, Cc, H = 2, 128, 16
    x = rb(torch.randn((N, Cc, H, H), generator=g))
    w1 = rb(torch.randn((32, Cc, 3, 3), generator=g) / np.sqrt(9 * Cc))
    b1 = torch.randn(32, generator=g) * 0.1
    w3 = rb(torch.randn((Cc, 32, 1, 1), generator=g) / np.sqrt(32))
    b3 = torch.randn(Cc, generator=g) * 0.1
    h_ref = F.relu(F.conv2d(F.relu(x), w1, b1, padding=1))
    hb = torch.empty((N, H, H, 32), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(x), packed_bf16(ops.pack_conv(w1.cuda())), b1.cuda(), hb, cin=Cc, cout=32, flags=ops.FO_IN_RELU | ops.FO_OUT_RELU)
    close_bf16(back(hb), h_ref, "resblock 3x3")
    out_ref = F.relu(F.conv2d(hb.float().cpu().permute(0, 3, 1, 2), w3, b3) + x)
    out = torch.empty((N, H, H, Cc), device="cuda", dtype=BF)
    ops.conv_bf16g(hb, packed_bf16(ops.pack_conv(w3.cuda())), b3.cuda(), out, k=(1, 1, 1), pad=(0, 0, 0), cin=32, cout=Cc, flags=ops.FO_OUT_RELU,
                   add=nhwc(x))
    close_bf16(back(out), out_ref, "resblock 1x1 + residual")
    # data gradients of the pair: 1x1 128 -> 32 with the hidden ReLU's mask, 3x3 32 -> 128 with the input ReLU's mask + the residual's gradient
    gy = rb(torch.randn((N, Cc, H, H), generator=g))
    gh_ref = F.conv_transpose2d(gy, w3) * (hb.float().cpu().permute(0, 3, 1, 2) > 0)
    gh = torch.empty((N, H, H, 32), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(gy), packed_bf16(ops.pack_conv_dgrad(w3.reshape(Cc, 32, 1).cuda())), None, gh, k=(1, 1, 1), pad=(0, 0, 0), cin=Cc, cout=32, mask=hb)
    close_bf16(back(gh), gh_ref, "resblock 1x1 data gradient")
    gx_ref = F.conv_transpose2d(gh.float().cpu().permute(0, 3, 1, 2), w1, padding=1) * (x > 0) + gy
    gx = torch.empty((N, H, H, Cc), device="cuda", dtype=BF)
    ops.conv_bf16g(gh, packed_bf16(ops.pack_conv_dgrad(w1.reshape(32, Cc, 9).cuda())), None, gx, cin=32, cout=Cc, mask=nhwc(x), add=nhwc(gy))
    close_bf16(back(gx), gx_ref, "resblock 3x3 data gradient")


def test_quantize_conv_1x1_192_channels_fp32_output():
    """quantize_conv_b (:213): 1x1, 192 -> 64 on the concatenated buffer, result kept in fp32 (FO_OUT_F32) for the quantiser"""
    from faceoff_amd import ops
    g = gen(5)
    N, H = 2, 24
    x = rb(torch.randn((N, 192, H, H), generator=g))
    w = rb(torch.randn((64, 192, 1, 1), generator=g) / np.sqrt(192))
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x, w, b)
    out = torch.empty((N, H, H, 64), device="cuda", dtype=torch.float32)
    ops.conv_bf16g(nhwc(x), packed_bf16(ops.pack_conv(w.cuda())), b.cuda(), out, k=(1, 1, 1), pad=(0, 0, 0), cin=192, cout=64)
    close_f32(back(out), ref, 2e-5, "quantize_conv_b")


def test_last_transposed_conv_as_one_cell_form_launch_fp32_output():
    """dec.blocks.6 (:152): ConvTranspose2d(64, 6, 4, 2, 1) as ONE launch (k2 conv over the cell grid, depth-to-space epilogue), fp32 result"""
    from faceoff_amd import ops
    g = gen(6)
    N, h = 2, 24
    x = rb(torch.randn((N, 64, h, h), generator=g))
    w = rb(torch.randn((64, 6, 4, 4), generator=g) / 16)
    b = torch.randn(6, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    out = torch.full((N, 2 * h, 2 * h, 8), 5.0, device="cuda", dtype=torch.float32)
    ops.convT_fused_bf16(nhwc(x), packed_bf16(ops.pack_convT_fused(w.cuda())), b.cuda(), out, cin=64, cout=6)
    close_f32(back(out)[:, :6], ref, 2e-5, "dec.blocks.6")
    assert bool((out[..., 6:] == 0).all())                  # the two padding channels are written as zeros
    # its data gradient: k4 s2 conv over the 8-channel gradient (16-byte pixels), 8 -> 64 with ReLU mask
    gy = torch.zeros((N, 8, 2 * h, 2 * h))
    gy[:, :6] = rb(torch.randn((N, 6, 2 * h, 2 * h), generator=g))
    mask = rb(torch.randn((N, 64, h, h), generator=g))
    gref = F.conv2d(gy[:, :6], w, stride=2, padding=1) * (mask > 0)
    gin = torch.empty((N, h, h, 64), device="cuda", dtype=BF)
    ops.conv_bf16g(nhwc(gy), packed_bf16(ops.pack_conv(w.cuda())), None, gin, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=8, cout=64, mask=nhwc(mask))
    close_bf16(back(gin), gref, "dec.blocks.6 data gradient")


# ------------------------------------------------------------------------------------------------ filter gradients
def _wgrad_case(seed, N, T, ca, cb, k, stride, pad, Hm, Wm, in_relu=False, ca_real=None, cb_real=None):
    from faceoff_amd import ops
    g = gen(seed)
    kd, kh, kw = k
    Hq, Wq = (Hm - 1) * stride + kh - 2 * pad[1], (Wm - 1) * stride + kw - 2 * pad[2]
    if stride == 2:
        Hq, Wq = 2 * Hm, 2 * Wm
    ca_real, cb_real = ca_real or ca, cb_real or cb
    P = torch.zeros((N, ca, Hm, Wm))
    P[:, :ca_real] = rb(torch.randn((N, ca_real, Hm, Wm), generator=g))
    Q = torch.zeros((N, cb, Hq, Wq))
    Q[:, :cb_real] = rb(torch.randn((N, cb_real, Hq, Wq), generator=g))
    Qe = F.relu(Q) if in_relu else Q
    w = torch.zeros((ca_real, cb_real, kd, kh, kw), requires_grad=True)
    B = N // T
    y = F.conv3d(Qe[:, :cb_real].reshape(B, T, cb_real, Hq, Wq).permute(0, 2, 1, 3, 4), w, stride=(1, stride, stride), padding=pad)
    y.backward(P[:, :ca_real].reshape(B, T, ca_real, Hm, Wm).permute(0, 2, 1, 3, 4))
    dw = torch.empty((ca_real, cb_real, kd * kh * kw), device="cuda")
    db = torch.empty(ca_real, device="cuda")
    ops.conv_wgrad_bf16(nhwc(P), nhwc(Q), dw, db, T=T, k=k, stride=stride, pad=pad, a_real=ca_real, b_real=cb_real, in_relu=in_relu)
    close_f32(dw.cpu().reshape(w.shape), w.grad, 2e-5, f"wgrad {ca}x{cb} k{k} s{stride} {Hm}x{Wm}")
    close_f32(db.cpu(), P[:, :ca_real].sum((0, 2, 3)), 2e-5, "bias gradient")


@pytest.mark.parametrize("T", [1, 2, 5])
def test_wgrad_conv3d_row_runs(T):
    _wgrad_case(20 + T, 2 * T, T, 128, 128, (3, 3, 3), 1, (1, 1, 1), 8, 32)


@pytest.mark.parametrize("ca,cb,in_relu", [(128, 128, False), (32, 128, True), (128, 64, False), (64, 128, False)])
def test_wgrad_3x3_row_runs(ca, cb, in_relu):
    """3x3 s1 layers: 128 -> 128, the ResBlocks' 128 -> 32 (its input ReLU applied as Q is staged), 64 -> 128 and its transpose"""
    _wgrad_case(30 + ca + cb, 3, 1, ca, cb, (1, 3, 3), 1, (0, 1, 1), 12, 64, in_relu=in_relu)


@pytest.mark.parametrize("ca,cb,k,T,N,Hm,Wm,in_relu", [
    (128, 128, (1, 3, 3), 1, 3, 12, 64, False),      # the 128 -> 128 layers
    (256, 64, (1, 3, 3), 1, 2, 5, 32, False),        # two tiles of P's channels, one of Q's
    (136, 96, (1, 3, 3), 1, 2, 6, 96, True),         # channel tails in both tiles (masked lanes), three runs per row, ReLU applied to the fragments
    (128, 192, (1, 3, 3), 1, 2, 1, 32, False),       # one-row images: both neighbour rows are padding
    (128, 128, (3, 3, 3), 5, 10, 8, 32, False),      # Conv3d: three depth planes, the outer ones skip a frame of each clip
    (128, 64, (3, 3, 3), 1, 3, 4, 32, False),        # Conv3d on one-frame clips: the outer planes have no work at all
    (128, 128, (1, 3, 3), 1, 40, 32, 64, False),     # enough units for several K-steps per slab on every CU (ring wrap-around)
    (32, 128, (1, 3, 3), 1, 3, 12, 64, True),        # the 32 x 128 block: the ResBlocks' 128 -> 32 with its input ReLU
    (24, 256, (1, 3, 3), 1, 2, 5, 32, False),        # ... a channel tail in P's tile, two tiles of Q's channels
    (32, 128, (3, 3, 3), 2, 4, 6, 32, True),         # ... as a Conv3d
    (32, 128, (1, 3, 3), 1, 48, 32, 64, True),       # ... several K-steps per slab
])
def test_wgrad_all_nine_taps_form(ca, cb, k, T, N, Hm, Wm, in_relu, monkeypatch):
    """wgrad9_bf16_kernel (3x3 pad-1 stride-1 filters between >= 128 and >= 64 channels, or <= 32 and >= 128: a workgroup holds all nine taps of a depth
    plane, LDS-DMA staging, swizzled tiles) against torch's fp32 filter gradient, and the row-run form it replaced (FACEOFF_WGRAD_ROWS=1) on the same operands: two summation orders
    of the same products.  The library reports which kernel ran."""
    from faceoff_amd import _lib
    lib = _lib.load()
    seed = 90 + ca + cb + k[0] + Hm
    lib.fo_kernel_notes(1); lib.fo_last_kernel()
    _wgrad_case(seed, N, T, ca, cb, k, 1, (k[0] // 2, 1, 1), Hm, Wm, in_relu=in_relu)
    torch.cuda.synchronize()
    kern = lib.fo_last_kernel().decode()
    assert "wgrad9_bf16_kernel" in kern, kern
    monkeypatch.setenv("FACEOFF_WGRAD_ROWS", "1")
    lib.fo_last_kernel()
    _wgrad_case(seed, N, T, ca, cb, k, 1, (k[0] // 2, 1, 1), Hm, Wm, in_relu=in_relu)
    torch.cuda.synchronize()
    kern = lib.fo_last_kernel().decode(); lib.fo_kernel_notes(0)
    assert "wgrad9" not in kern and "wgrad_bf16_kernel" in kern, kern


def test_wgrad_all_nine_taps_form_equals_the_row_run_form_on_random_shapes(monkeypatch):
    """Sixty seeded random geometries of the two wgrad9 block shapes -- channel tails, one- to 32-row frames, one to four runs per row, Conv3d with clips of
    1..5 frames, with and without the input ReLU -- on the same operands through both forms: the same products in another order, so the filter and bias gradients agree to
    summation-order noise (2e-5 of the largest entry; measured <= 1e-7 over 120 shapes)."""
    import random
    from faceoff_amd import ops
    rnd = random.Random(12345)
    for it in range(60):
        thin = rnd.random() < 0.35
        ca = rnd.choice([8, 24, 32]) if thin else rnd.choice([128, 136, 192, 256])
        cb = rnd.choice([128, 256] if thin else [64, 96, 128, 192])
        kd = rnd.choice([1, 1, 3])
        T = rnd.choice([1, 2, 3, 5]) if kd == 3 else 1
        N = rnd.choice([1, 2, 3, 7]) * T
        Hm, Wm = rnd.choice([1, 2, 3, 5, 8, 13, 32]), rnd.choice([32, 64, 96, 128])
        relu = rnd.random() < 0.4
        g = torch.Generator(device="cuda").manual_seed(it)
        P = torch.zeros((N, Hm, Wm, (ca + 7) // 8 * 8), device="cuda", dtype=BF)
        P[..., :ca] = torch.randn((N, Hm, Wm, ca), device="cuda", generator=g).to(BF)
        Q = torch.randn((N, Hm, Wm, cb), device="cuda", generator=g).to(BF)
        res = []
        for rows in (False, True):
            if rows:
                monkeypatch.setenv("FACEOFF_WGRAD_ROWS", "1")
            else:
                monkeypatch.delenv("FACEOFF_WGRAD_ROWS", raising=False)
            dw = torch.full((ca, cb, kd * 9), 7.0, device="cuda")
            db = torch.full((ca,), 7.0, device="cuda")
            ops.conv_wgrad_bf16(P, Q, dw, db, T=T, k=(kd, 3, 3), stride=1, pad=(kd // 2, 1, 1), a_real=ca, b_real=cb, in_relu=relu)
            res.append((dw, db))
        what = f"ca={ca} cb={cb} kd={kd} T={T} N={N} {Hm}x{Wm} relu={relu}"
        assert bool(torch.isfinite(res[0][0]).all()), what
        close_f32(res[0][0], res[1][0], 2e-5, "filter gradient, " + what)
        close_f32(res[0][1], res[1][1], 2e-5, "bias gradient, " + what)
    monkeypatch.delenv("FACEOFF_WGRAD_ROWS", raising=False)


@pytest.mark.parametrize("ca,cb", [(128, 32), (64, 128), (64, 192)])
def test_wgrad_1x1(ca, cb):
    _wgrad_case(40 + ca + cb, 2, 1, ca, cb, (1, 1, 1), 1, (0, 0, 0), 16, 32)


@pytest.mark.parametrize("ca,cb", [(128, 64), (64, 128), (64, 64)])
def test_wgrad_k4s2_row_runs(ca, cb):
    """k4 s2 stems (and, with the roles of input and output gradient swapped, the transposed convs): four kw taps per staged row"""
    _wgrad_case(50 + ca + cb, 2, 1, ca, cb, (1, 4, 4), 2, (0, 1, 1), 10, 32)


def test_wgrad_image_layers_8_channels():
    """enc_b.blocks.0 / dec.blocks.6: Q is the 8-channel (6 real) image-side tensor, k4 s2; the four kw taps are the b index"""
    _wgrad_case(60, 2, 1, 64, 8, (1, 4, 4), 2, (0, 1, 1), 16, 64, cb_real=6)


@pytest.mark.parametrize("k,stride,pad,ca,cb,T", [((1, 3, 3), 1, (0, 1, 1), 128, 128, 1), ((3, 3, 3), 1, (1, 1, 1), 128, 128, 3), ((1, 4, 4), 2, (0, 1, 1), 128, 64, 1),
                                                    ((1, 1, 1), 1, (0, 0, 0), 128, 32, 1), ((1, 3, 3), 1, (0, 1, 1), 32, 128, 1)])
def test_wgrad_gather_form_for_other_widths(k, stride, pad, ca, cb, T):
    """widths that are not multiples of 32 (40 x 24 frames and the like): one tap per workgroup, every row's source pixel decoded on its own"""
    _wgrad_case(70 + ca + cb + k[2], 2 * T, T, ca, cb, k, stride, pad, 10, 12, in_relu=(ca == 32))


# ------------------------------------------------------------------------------------------------ element-wise
def test_conversions_input_layout_and_vq_glue():
    from faceoff_amd import ops
    g = gen(9)
    x = torch.randn((3, 5, 7, 64), generator=g).cuda()
    wide = torch.zeros((3, 5, 7, 192), device="cuda", dtype=BF)
    ops.to_bf16(x, wide[..., 64:128])
    assert torch.equal(wide[..., 64:128], x.to(BF)) and bool((wide[..., :64] == 0).all()) and bool((wide[..., 128:] == 0).all())
    assert torch.equal(ops.to_f32(wide[..., 64:128]), x.to(BF).float())
    a, b = torch.randn((2, 3, 6, 10), generator=g).cuda(), torch.randn((2, 3, 6, 10), generator=g).cuda()
    y = ops.cat_nchw_to_nhwc8_bf16(a, b)
    want = torch.zeros((2, 6, 10, 8), device="cuda")
    want[..., :3], want[..., 3:6] = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
    assert torch.equal(y, want.to(BF))
    # quantiser glue: fo_vq_assign2 = fo_vq_assign + a bf16 copy; fo_vq_bwd_bf16 = bf16(gq + gdiff * 2/numel * (x - q))
    xin = (torch.randn((2, 4, 4, 64), generator=g) * 0.3).cuda()
    embed = (torch.randn((64, 512), generator=g) * 0.3).cuda()
    eT, en = ops.vq_prepare(embed)
    q1, st1 = torch.empty_like(xin), torch.zeros(1 + 512 + 512 * 64, device="cuda")
    i1 = ops.vq_assign(xin, eT, en, q1, st1, True)
    q2, st2 = torch.empty_like(xin), torch.zeros(1 + 512 + 512 * 64, device="cuda")
    qb = torch.zeros((2, 4, 4, 128), device="cuda", dtype=BF)
    i2 = ops.vq_assign_bf16out(xin, eT, en, q2, qb[..., 64:], st2, True)
    assert torch.equal(i1, i2) and torch.equal(q1, q2) and torch.equal(st1[1:], st2[1:]) and torch.equal(qb[..., 64:], q1.to(BF)) and bool((qb[..., :64] == 0).all())
    np.testing.assert_allclose(st1[0].item(), st2[0].item(), rtol=1e-6)
    gq = torch.randn((2, 4, 4, 64), generator=g).cuda().to(BF)
    gd = torch.tensor([0.7], device="cuda")
    gx = torch.empty((2, 4, 4, 64), device="cuda", dtype=BF)
    ops.vq_bwd_bf16(gq, xin, q1, gd, gx)
    want = (gq.float() + 0.7 * (2.0 / xin.numel()) * (xin - q1)).to(BF)
    assert (gx.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()


@pytest.mark.parametrize("N,H,W", [(3, 16, 32), (2, 6, 64), (1, 32, 32)])
def test_resblock_conv1_halo_tile_kernel_bf16_vs_torch_and_vs_the_tiled_kernel(N, H, W):
    """conv3x3_c128to32_halo_bf16_kernel (csrc/resblock_bf16.hip: ReLU -> Conv2d(128, 32, 3, padding=1) -> ReLU of a ResBlock, reference
    models/vqvae_conv3d_latent.py:91-93, on bf16 operands; taken at C3 sizes, forced here) against the same arithmetic in torch (bf16-rounded
    operands, wide accumulation, one rounding of the result) and against the tiled kernel: both within one bf16 rounding of the exact value,
    and equal to each other except where fp32 summation order moves a result across a rounding boundary."""
    import subprocess, sys, json
    code = r"""
import sys, json, torch
sys.path.insert(0, %r)
from faceoff_amd import ops
N, H, W = %d, %d, %d
g = torch.Generator().manual_seed(N * 100 + H)
x = torch.randn((N, 128, H, W), generator=g).to(torch.bfloat16)
w = (torch.randn((32, 128, 3, 3), generator=g) * 0.05).to(torch.bfloat16)
b = torch.randn(32, generator=g) * 0.1
ref = torch.relu(torch.nn.functional.conv2d(torch.relu(x.double()), w.double(), b.double(), padding=1))
xc = x.permute(0, 2, 3, 1).contiguous().cuda()
wide = torch.zeros((N, H, W, 192), device="cuda", dtype=torch.bfloat16); wide[..., 64:192] = xc
wp = ops.to_bf16(ops.pack_conv(w.float().cuda()))
res = {}
for name, xin in (("dense", xc), ("slice", wide[..., 64:192])):
    out = torch.full((N, H, W, 48), 5.0, device="cuda", dtype=torch.bfloat16)
    ops.conv_bf16g(xin, wp, b.cuda(), out[..., 8:40], cin=128, cout=32, flags=ops.FO_IN_RELU | ops.FO_OUT_RELU, k=(1, 3, 3), stride=1, pad=(0, 1, 1))
    assert (out[..., :8] == 5.0).all() and (out[..., 40:] == 5.0).all()
    y = out[..., 8:40].permute(0, 3, 1, 2).float().cpu().double()
    res[name] = {"err": ((y - ref).abs() / (ref.abs() + 1.0)).max().item(), "y": y.flatten()[::7].tolist()}
# the same convolution backwards (conv3x3_c32to128_halo_bf16_kernel): g_x = conv3x3^T(g_h) * (x > 0) + g_out
gh = torch.randn((N, 32, H, W), generator=g).to(torch.bfloat16); go = torch.randn((N, 128, H, W), generator=g).to(torch.bfloat16)
gref = torch.nn.functional.conv_transpose2d(gh.double(), w.double(), padding=1) * (x.double() > 0) + go.double()
wpd = ops.to_bf16(ops.pack_conv_dgrad(w.float().cuda().reshape(32, 128, -1)))
ghc, goc = gh.permute(0, 2, 3, 1).contiguous().cuda(), go.permute(0, 2, 3, 1).contiguous().cuda()
wg = torch.zeros((N, H, W, 64), device="cuda", dtype=torch.bfloat16); wg[..., 16:48] = ghc
for name, gin, xin in (("dgrad dense", ghc, xc), ("dgrad slice", wg[..., 16:48], wide[..., 64:192])):
    gx = torch.full((N, H, W, 128), 9.0, device="cuda", dtype=torch.bfloat16)
    ops.conv_bf16g(gin, wpd, None, gx, cin=32, cout=128, mask=xin, add=goc, k=(1, 3, 3), stride=1, pad=(0, 1, 1))
    y = gx.permute(0, 3, 1, 2).float().cpu().double()
    res[name] = {"err": ((y - gref).abs() / (gref.abs() + 1.0)).max().item(), "y": y.flatten()[::7].tolist()}
print(json.dumps(res))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), N, H, W)
    outs = {}
    for mode in ("halo", "tiled"):       # (the switch is read once per process)
        env = dict(os.environ, **({"FACEOFF_FORCE_RESBLOCK_HALO": "1"} if mode == "halo" else {"FACEOFF_NO_RESBLOCK_HALO": "1"}))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for mode, res in outs.items():
        for name, v in res.items():
            assert v["err"] <= 2.0 ** -8, (mode, name, v["err"])          # one bf16 rounding (2^-9 relative) + accumulation noise
    for key in ("dense", "dgrad dense"):
        a, b = np.array(outs["halo"][key]["y"]), np.array(outs["tiled"][key]["y"])
        assert np.array_equal(np.array(outs["halo"][key.replace("dense", "slice")]["y"]), a)         # the pixel stride does not enter the arithmetic
        assert (a != b).mean() < 0.02 and np.abs(a - b).max() <= 2.0 ** -7 * (np.abs(b).max() + 1.0), key


# ------------------------------------------------------------------------------------------------ the extended-tile (halo) ping-pong kernel
@pytest.mark.parametrize("N,H,W,cin,cout,T,tile512", [
    (4, 16, 16, 256, 256, 1, False),      # 256 x 256 tile, one frame per tile, 3 pieces per wave
    (2, 32, 32, 128, 256, 1, False),      # 256 x 256, 8 image rows per tile
    (1, 64, 64, 64, 256, 1, False),       # 256 x 256, W = 64 (4 image rows per tile); 64 input channels = 2 chunks
    (1, 128, 128, 128, 128, 1, True),     # 512 x 128 tile, W = 128: 6 pieces per wave (VGG conv2_2's shape)
    (2, 64, 64, 128, 128, 1, True),       # 512 x 128, W = 64: 5 pieces per wave
    (6, 32, 32, 128, 128, 3, True),       # 512 x 128 with depth taps, clips of 3 frames (the VQ-VAE's Conv3d at the top latent size)
    (5, 32, 32, 128, 128, 5, False),      # 256 x 128 with depth taps, one clip of 5
    (2, 32, 64, 96, 128, 1, False),       # 256 x 128, a non-square frame, 96 input channels (3 chunks)
])
def test_extended_tile_kernel_vs_torch_and_vs_the_per_tap_kernel(N, H, W, cin, cout, T, tile512, monkeypatch):
    """conv_bf16_pph_kernel (the A operand staged once per (depth tap, 32-channel chunk) for the nine 3x3 taps: kw shifts read the staged
    rows at +-1 with a zero row for lanes that would wrap into the neighbouring image row, kh shifts at +-W) at every tile shape and piece
    count it is built for: forward (bias, ReLU) and masked data gradient with fan-in add against torch-CPU on the same bf16 operands, and
    against conv_bf16_pp16_kernel (FACEOFF_BF16_NO_PPH=1) -- same products, another summation order: equal up to isolated 1-ulp roundings.
    The library reports which kernel ran (fo_last_kernel)."""
    from faceoff_amd import ops, _lib
    monkeypatch.setenv("FACEOFF_BF16_BIG_TILES", "1")
    if tile512:
        monkeypatch.setenv("FACEOFF_BF16_TILE512", "1")
    g = gen(N * 1000 + W + cin)
    kd = 3 if T > 1 else 1
    x = rb(torch.randn((N, cin, H, W), generator=g))
    w = rb(torch.randn((cout, cin, kd, 3, 3), generator=g) / np.sqrt(9 * kd * cin))
    b = torch.randn(cout, generator=g)
    mask = rb(torch.randn((N, cout, H, W), generator=g))
    add = rb(torch.randn((N, cout, H, W), generator=g))
    B = N // T
    if kd == 3:
        x5 = x.reshape(B, T, cin, H, W).permute(0, 2, 1, 3, 4)
        ref = F.conv3d(x5, w, b, padding=1).permute(0, 2, 1, 3, 4).reshape(N, cout, H, W)
    else:
        ref = F.conv2d(x, w[:, :, 0], b, padding=1)
    ref_fwd = F.relu(ref)
    ref_msk = (ref - b.view(1, -1, 1, 1)) * (mask > 0) + add
    wp = packed_bf16(ops.pack_conv(w.reshape(cout, cin, kd * 9).cuda()))
    xin = torch.zeros((N, H, W, cin + 32), device="cuda", dtype=BF)          # a channel slice of a wider buffer (ldIn > Cin)
    xin[..., :cin] = nhwc(x)
    lib = _lib.load()
    outs = {}
    for name, env in (("pph", "0"), ("pp16", "1")):
        monkeypatch.setenv("FACEOFF_BF16_NO_PPH", env)
        o1 = torch.empty((N, H, W, cout), device="cuda", dtype=BF)
        lib.fo_kernel_notes(1); lib.fo_last_kernel()
        ops.conv_bf16g(xin[..., :cin], wp, b.cuda(), o1, T=T, k=(kd, 3, 3), pad=(kd // 2, 1, 1), cin=cin, cout=cout, flags=ops.FO_OUT_RELU)
        kern = lib.fo_last_kernel().decode(); lib.fo_kernel_notes(0)
        assert kern.startswith("conv_bf16_pph_kernel<" if name == "pph" else "conv_bf16_pp16_kernel<"), kern
        want_tile = "512, 128" if tile512 else ("256, 256" if cout % 256 == 0 else "256, 128")
        assert want_tile in kern, kern
        o2 = torch.empty((N, H, W, cout), device="cuda", dtype=BF)
        ops.conv_bf16g(xin[..., :cin], wp, None, o2, T=T, k=(kd, 3, 3), pad=(kd // 2, 1, 1), cin=cin, cout=cout, mask=nhwc(mask), add=nhwc(add))
        torch.cuda.synchronize()
        close_bf16(back(o1), ref_fwd, f"{name} forward {kern}")
        close_bf16(back(o2), ref_msk, f"{name} masked + add {kern}")
        outs[name] = (o1.float(), o2.float())
    for a_, b_ in zip(outs["pph"], outs["pp16"]):
        diff = (a_ - b_).abs()
        assert (diff > 0).float().mean().item() < 2e-2 and bool((diff <= b_.abs() * 2.0 ** -7 + 1e-3 * b_.abs().max()).all())


@pytest.mark.parametrize("N,H,W,cin,cout,T,tile512", [
    (20, 64, 64, 128, 256, 1, False),     # 320 tiles of 256 x 256 on 256 workgroups: some walk two tiles, some one
    (36, 64, 64, 128, 128, 3, True),      # 288 tiles of 512 x 128, clips of 3: the tiles of a clip's first / last frame have fewer K groups
])
def test_extended_tile_kernel_persistent_grid_equals_one_workgroup_per_tile(N, H, W, cin, cout, T, tile512, monkeypatch):
    """FACEOFF_BF16_PPH_PERSIST=1: one workgroup per CU walks several tiles, the next tile's first operands in flight while this tile is stored.
    Same arithmetic as one workgroup per tile, so the results must be BIT-equal -- in every form of the epilogue: bf16 through the LDS patch
    (forward; masked data gradient) and from the accumulators (fan-in add; fp32 output)."""
    from faceoff_amd import ops, _lib
    if tile512:
        monkeypatch.setenv("FACEOFF_BF16_TILE512", "1")
    g = gen(N + W + cin)
    kd = 3 if T > 1 else 1
    x = (torch.randn((N, H, W, cin), generator=g) * 0.5).to(BF).cuda()
    w = torch.randn((cout, cin, kd * 9), generator=g) / np.sqrt(9 * kd * cin)
    wp = packed_bf16(ops.pack_conv(w.cuda()))
    b = torch.randn(cout, generator=g).cuda()
    mask = torch.randn((N, H, W, cout), generator=g).to(BF).cuda()
    add = torch.randn((N, H, W, cout), generator=g).to(BF).cuda()
    lib = _lib.load()
    kw = dict(T=T, k=(kd, 3, 3), pad=(kd // 2, 1, 1), cin=cin, cout=cout)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FACEOFF_BF16_PPH_PERSIST", mode)
        outs = [torch.empty((N, H, W, cout), device="cuda", dtype=BF) for _ in range(3)] + [torch.empty((N, H, W, cout), device="cuda", dtype=torch.float32)]
        lib.fo_kernel_notes(1); lib.fo_last_kernel()
        ops.conv_bf16g(x, wp, b, outs[0], flags=ops.FO_OUT_RELU, **kw)
        kern = lib.fo_last_kernel().decode(); lib.fo_kernel_notes(0)
        assert kern.startswith("conv_bf16_pph_kernel<"), kern
        ops.conv_bf16g(x, wp, None, outs[1], mask=mask, **kw)
        ops.conv_bf16g(x, wp, None, outs[2], mask=mask, add=add, **kw)
        ops.conv_bf16g(x, wp, b, outs[3], **kw)
        if T == 1:                                       # the bit-plane forms (fo_conv_bf16_ex): the result's sign plane, the mask from a plane
            plane = torch.empty((N, H, W, cout // 8), device="cuda", dtype=torch.uint8)
            o5, o6 = torch.empty_like(outs[0]), torch.empty_like(outs[0])
            ops.conv_bf16(x, wp, b, o5, cin=cin, cout=cout, flags=ops.FO_OUT_RELU, out_bits=plane)
            mplane = ((mask.float() > 0).reshape(N, H, W, cout // 8, 8).long() << torch.arange(8, device="cuda")).sum(-1).to(torch.uint8)
            ops.conv_bf16(x, wp, None, o6, cin=cin, cout=cout, mask_bits=mplane)
            outs += [o5, o6, plane]
        torch.cuda.synchronize()
        res[mode] = outs
    for a_, b_ in zip(res["0"], res["1"]):
        assert torch.equal(a_, b_)
    if T == 1:
        assert torch.equal(res["1"][4], res["1"][0]) and torch.equal(res["1"][5], res["1"][1])       # planes change nothing in the values
    # and the values themselves: against the per-tap kernel
    monkeypatch.setenv("FACEOFF_BF16_NO_PPH", "1")
    ref = torch.empty((N, H, W, cout), device="cuda", dtype=BF)
    ops.conv_bf16g(x, wp, None, ref, mask=mask, add=add, **kw)
    torch.cuda.synchronize()
    diff = (res["1"][2].float() - ref.float()).abs()
    assert (diff > 0).float().mean().item() < 2e-2 and bool((diff <= ref.float().abs() * 2.0 ** -7 + 1e-3 * ref.float().abs().max()).all())

