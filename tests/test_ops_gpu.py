"""Per-kernel parity on a real MI355X: every C-ABI op against the CPU oracle (torch fp32 CPU for the
floating-point convs -- the reference's own arithmetic -- and oracle/vq_oracle.c for the VQ search).
Tolerance: 1e-3 relative (BASELINE.json north_star), measured against the tensor's scale; observed
errors are ~1e-6.  VQ indices: bit-exact."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    return torch.device("cuda:0")


def _close(got, want, rtol=RTOL, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"
    return err / scale


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def _nhwc(t):   # NCHW cpu -> NHWC cuda (padded to 8 channels if fewer)
    t = t.permute(0, 2, 3, 1).contiguous()
    if t.shape[-1] < 8:
        t = F.pad(t, (0, 8 - t.shape[-1]))
    return t.to(_dev())


CONV_CASES = [
    # (name, Cin, Cout, k, stride, N, H, W)
    ("enc_b0_6to64_k4s2", 6, 64, 4, 2, 2, 32, 32),
    ("k4s2_64to128", 64, 128, 4, 2, 3, 16, 24),
    ("k4s2_128to64", 128, 64, 4, 2, 2, 16, 16),
    ("k3_128to128", 128, 128, 3, 1, 2, 16, 16),
    ("k3_64to128_odd", 64, 128, 3, 1, 3, 10, 14),
    ("k3_128to32", 128, 32, 3, 1, 2, 16, 16),
    ("k1_32to128", 32, 128, 1, 1, 2, 16, 16),
    ("k1_128to64", 128, 64, 1, 1, 2, 8, 8),
    ("k1_192to64", 192, 64, 1, 1, 2, 16, 16),
    # widths that are multiples of 32: the wgrad FASTROW path (scalar row masks) and full 128-pixel tiles
    ("k3_128to128_w32", 128, 128, 3, 1, 1, 8, 32),
    ("k3_128to32_w64", 128, 32, 3, 1, 1, 6, 64),
    ("k4s2_64to128_w64", 64, 128, 4, 2, 2, 8, 64),
    ("k4s2_128to64_w128", 128, 64, 4, 2, 1, 4, 128),
    ("k1_32to128_w32", 32, 128, 1, 1, 2, 4, 32),
    ("enc_b0_6to64_w64", 6, 64, 4, 2, 1, 8, 64),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd_dgrad_wgrad(case):
    from faceoff_amd import ops
    name, Ci, Co, k, s, N, H, W = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = _rand(rng, N, Ci, H, W).requires_grad_(True)
    w = _rand(rng, Co, Ci, k, k, scale=0.1).requires_grad_(True)
    b = _rand(rng, Co, scale=0.1).requires_grad_(True)
    pad = 1 if k > 1 else 0
    y = F.conv2d(x, w, b, stride=s, padding=pad)
    gy = _rand(rng, *y.shape)
    y.backward(gy)
    dev = _dev()
    xg, wg, bg = _nhwc(x.detach()), w.detach().to(dev), b.detach().to(dev)
    out = torch.empty((N, y.shape[2], y.shape[3], Co), device=dev)
    wp = ops.pack_conv(wg)
    ops.conv_igemm(xg, wp, bg, out, k=(1, k, k), stride=s, pad=(0, pad, pad), cin=ops.pad_in(Ci), cout=Co)
    _close(out.permute(0, 3, 1, 2), y, what="fwd")
    # wgrad (+ bias grad riding along)
    gyg = _nhwc(gy)
    dw, db = torch.empty_like(wg), torch.empty_like(bg)
    ops.conv_wgrad(gyg, xg, dw, db, k=(1, k, k), stride=s, pad=(0, pad, pad), a_real=Co, b_real=Ci)
    _close(dw, w.grad, what="wgrad")
    _close(db, b.grad, what="bias grad")
    if Ci < 32:
        return
    gx = torch.empty((N, H, W, Ci), device=dev)
    if k == 4:
        wpd = ops.pack_convT(wg)
        ops.convT_phases(gyg, wpd, None, gx, cin=Co, cout=Ci)
    else:
        wpd = ops.pack_conv_dgrad(wg.reshape(Co, Ci, -1))
        ops.conv_igemm(gyg, wpd, None, gx, k=(1, k, k), stride=1, pad=(0, k - 1 - pad, k - 1 - pad), cin=Co, cout=Ci)
    _close(gx.permute(0, 3, 1, 2), x.grad, what="dgrad")


def test_conv_epilogue_flags_and_views():
    """ResBlock pieces: IN_RELU, OUT_RELU, residual add, mask, and channel-slice views (free torch.cat)."""
    from faceoff_amd import ops
    from faceoff_amd.ops import FO_IN_RELU, FO_OUT_RELU
    rng = np.random.default_rng(5)
    dev = _dev()
    N, H, W = 2, 16, 16
    x = _rand(rng, N, 128, H, W)
    w1, b1 = _rand(rng, 32, 128, 3, 3, scale=0.05), _rand(rng, 32, scale=0.1)
    w3, b3 = _rand(rng, 128, 32, 1, 1, scale=0.1), _rand(rng, 128, scale=0.1)
    h_ref = F.relu(F.conv2d(F.relu(x), w1, b1, padding=1))
    y_ref = F.relu(F.conv2d(h_ref, w3, b3) + x)
    xg = _nhwc(x)
    wide = torch.zeros((N, H, W, 192), device=dev)          # write the block output into a slice of a wider buffer
    h = torch.empty((N, H, W, 32), device=dev)
    ops.conv_igemm(xg, ops.pack_conv(w1.to(dev)), b1.to(dev), h, k=(1, 3, 3), pad=(0, 1, 1), cin=128, cout=32,
                   flags=FO_IN_RELU | FO_OUT_RELU)
    ops.conv_igemm(h, ops.pack_conv(w3.to(dev)), b3.to(dev), wide[..., 64:192], k=(1, 1, 1), pad=(0, 0, 0), cin=32, cout=128,
                   flags=FO_OUT_RELU, add=xg)
    _close(h.permute(0, 3, 1, 2), h_ref, what="relu-conv-relu")
    _close(wide[..., 64:192].permute(0, 3, 1, 2), y_ref, what="1x1 + residual + relu into slice")
    assert wide[..., :64].abs().max().item() == 0.0
    # masked dgrad with fan-in add: g_x = g + dgrad(gh) * (x > 0)
    gh = _rand(rng, N, 32, H, W)
    g = _rand(rng, N, 128, H, W)
    want = g + F.conv_transpose2d(gh, w1, padding=1) * (x > 0)
    gx = torch.empty((N, H, W, 128), device=dev)
    ops.conv_igemm(_nhwc(gh), ops.pack_conv_dgrad(w1.to(dev).reshape(32, 128, -1)), None, gx, k=(1, 3, 3), pad=(0, 1, 1),
                   cin=32, cout=128, mask=xg, add=_nhwc(g))
    _close(gx.permute(0, 3, 1, 2), want, what="masked dgrad + add")
    # wgrad with IN_RELU on Q
    xr = x.clone().requires_grad_(False)
    w1r = w1.clone().requires_grad_(True)
    F.conv2d(F.relu(xr), w1r, None, padding=1).backward(gh)
    dw = torch.empty_like(w1, device=dev)
    ops.conv_wgrad(_nhwc(gh), xg, dw, None, k=(1, 3, 3), pad=(0, 1, 1), a_real=32, b_real=128, in_relu=True)
    _close(dw, w1r.grad, what="wgrad in_relu")


CONVT_CASES = [("convT_128to64", 128, 64, 2, 8, 8), ("convT_64to64", 64, 64, 2, 8, 12), ("convT_64to6", 64, 6, 2, 16, 16),
               ("convT_128to64_w32", 128, 64, 1, 4, 32), ("convT_64to6_w64", 64, 6, 1, 4, 64)]


@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_conv_transpose_fwd_dgrad_wgrad(case):
    from faceoff_amd import ops
    name, Ci, Co, N, H, W = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = _rand(rng, N, Ci, H, W).requires_grad_(True)
    w = _rand(rng, Ci, Co, 4, 4, scale=0.1).requires_grad_(True)
    b = _rand(rng, Co, scale=0.1).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    gy = _rand(rng, *y.shape)
    y.backward(gy)
    dev = _dev()
    xg, wg, bg = _nhwc(x.detach()), w.detach().to(dev), b.detach().to(dev)
    Cs = max(Co, 8)
    out = torch.zeros((N, 2 * H, 2 * W, Cs), device=dev)
    ops.convT_phases(xg, ops.pack_convT(wg), bg, out, cin=Ci, cout=Co)
    _close(out[..., :Co].permute(0, 3, 1, 2), y, what="convT fwd")
    if Co <= 8:   # the fused single-launch form (4 phases x 8 channels as GEMM columns, depth-to-space epilogue)
        out2 = torch.zeros((N, 2 * H, 2 * W, 8), device=dev)
        ops.convT_fused(xg, ops.pack_convT_fused(wg), bg, out2, cin=Ci, cout=Co)
        _close(out2[..., :Co].permute(0, 3, 1, 2), y, what="convT fused fwd")
        assert out2[..., Co:].abs().max().item() == 0.0
    gyg = _nhwc(gy)
    gx = torch.empty((N, H, W, Ci), device=dev)
    ops.conv_igemm(gyg, ops.pack_conv(wg), None, gx, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=ops.pad_in(Co), cout=Ci)
    _close(gx.permute(0, 3, 1, 2), x.grad, what="convT dgrad")
    dw, db = torch.empty_like(wg), torch.empty_like(bg)
    ops.conv_wgrad(xg, gyg, dw, None, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=Ci, b_real=Co)
    ops.bias_grad(gyg, db, Co)
    _close(dw, w.grad, what="convT wgrad")
    _close(db, b.grad, what="convT bias grad")


@pytest.mark.parametrize("B,T,H,W", [(1, 1, 8, 16), (2, 2, 8, 8), (1, 5, 16, 16), (2, 5, 16, 8), (3, 3, 6, 10),
                                     (2, 5, 4, 32), (1, 3, 2, 64), (2, 2, 8, 32)])
def test_conv3d_fwd_dgrad_wgrad(B, T, H, W):
    """Conv3d 128->128 k3 p1 (Conv3dLatentPostnet :181,185) incl. T=1,2,5 and tiles that straddle frames."""
    from faceoff_amd import ops
    rng = np.random.default_rng(100 + B * 10 + T)
    x = _rand(rng, B, 128, T, H, W).requires_grad_(True)
    w = _rand(rng, 128, 128, 3, 3, 3, scale=0.02).requires_grad_(True)
    b = _rand(rng, 128, scale=0.1).requires_grad_(True)
    y = F.conv3d(x, w, b, padding=1)
    gy = _rand(rng, *y.shape)
    y.backward(gy)
    dev = _dev()

    def frames(t5):   # [B,C,T,H,W] -> [B*T,H,W,C]
        return t5.detach().permute(0, 2, 3, 4, 1).reshape(B * T, H, W, 128).contiguous().to(dev)

    def clips(t4):    # inverse
        return t4.reshape(B, T, H, W, 128).permute(0, 4, 1, 2, 3).cpu()

    xg, gyg, wg, bg = frames(x), frames(gy), w.detach().to(dev), b.detach().to(dev)
    out = torch.empty_like(xg)
    ops.conv_igemm(xg, ops.pack_conv(wg), bg, out, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=128, cout=128)
    _close(clips(out), y, what="conv3d fwd")
    gx = torch.empty_like(xg)
    ops.conv_igemm(gyg, ops.pack_conv_dgrad(wg.reshape(128, 128, -1)), None, gx, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=128,
                   cout=128)
    _close(clips(gx), x.grad, what="conv3d dgrad")
    dw, db = torch.empty_like(wg), torch.empty_like(bg)
    ops.conv_wgrad(gyg, xg, dw, db, T=T, k=(3, 3, 3), pad=(1, 1, 1), a_real=128, b_real=128)
    _close(dw, w.grad, what="conv3d wgrad")
    _close(db, b.grad, what="conv3d bias grad")


@pytest.mark.parametrize("B,T,H,W,m", [(1, 1, 8, 16, 2), (2, 2, 8, 8, 2), (1, 5, 16, 16, 2), (2, 5, 32, 32, 2), (3, 3, 6, 10, 2),
                                       (1, 5, 16, 32, 2), (2, 2, 16, 16, 2), (1, 5, 64, 64, 4), (2, 2, 32, 64, 4), (1, 3, 16, 16, 4),
                                       (1, 1, 8, 12, 4)])
def test_conv3d_winograd_fwd_dgrad(B, T, H, W, m):
    """The Winograd F(m x m, 3x3) forms (m = 2, 4) of Conv3d 128->128 k3 p1 (transforms + banked (3,1,1) implicit GEMM)
    against torch-CPU: forward with bias + ReLU, data gradient with ReLU mask + residual, filter gradient; banked
    (tiles per frame % 128 == 0) and per-plane launches, clip padding (T = 1, 3, 5), a channel-slice output view."""
    from faceoff_amd import ops
    rng = np.random.default_rng(300 + B * 10 + T + H + m)
    x = _rand(rng, B, 128, T, H, W).requires_grad_(True)
    w = _rand(rng, 128, 128, 3, 3, 3, scale=0.02)
    b = _rand(rng, 128, scale=0.1)
    y_pre = F.conv3d(x, w, b, padding=1)
    gy = _rand(rng, *y_pre.shape)
    y_pre.backward(gy)
    dev = _dev()

    def frames(t5):   # [B,C,T,H,W] -> [B*T,H,W,C]
        return t5.detach().permute(0, 2, 3, 4, 1).reshape(B * T, H, W, 128).contiguous().to(dev)

    def clips(t4):
        return t4.reshape(B, T, H, W, 128).permute(0, 4, 1, 2, 3).cpu()

    xg, gyg, wg, bg = frames(x), frames(gy), w.to(dev), b.to(dev)
    wide = torch.zeros((B * T, H, W, 192), device=dev)
    ops.conv3d_winograd(xg, ops.wino_filter(wg, m=m), bg, wide[..., 64:192], T=T, cin=128, cout=128, flags=ops.FO_OUT_RELU, m=m)
    tol = 2e-5 if m == 2 else 2e-4       # F(2x2): the direct convolution's rounding; F(4x4): ~10x that (vs fp32 torch-CPU)
    _close(clips(wide[..., 64:192].contiguous()), torch.relu(y_pre), rtol=tol, what="winograd conv3d fwd")
    assert (wide[..., :64] == 0).all()
    mask = frames(_rand(rng, B, 128, T, H, W)).clamp_min(0)
    addt = frames(_rand(rng, B, 128, T, H, W))
    gx = torch.empty_like(xg)
    ops.conv3d_winograd(gyg, ops.wino_filter(wg, dgrad=True, m=m), None, gx, T=T, cin=128, cout=128, mask=mask, add=addt, m=m)
    want = x.grad * (clips(mask) > 0) + clips(addt)
    _close(clips(gx), want, rtol=tol, what="winograd conv3d dgrad")
    if ops.wino_wgrad_ok(H, W, B * T, T, m):             # filter gradient in the transformed domain (banked wgrad GEMMs)
        w2 = w.clone().requires_grad_(True)
        b2 = b.clone().requires_grad_(True)
        F.conv3d(x.detach(), w2, b2, padding=1).backward(gy)
        dw, db = torch.empty_like(wg), torch.empty_like(bg)
        ops.conv3d_wgrad_winograd(gyg, xg, dw, db, T=T, a_real=128, b_real=128, m=m)
        _close(dw, w2.grad, rtol=tol, what="winograd conv3d wgrad")
        _close(db, b2.grad, rtol=2e-5, what="winograd conv3d bias grad")


@pytest.mark.parametrize("N,H,W,m", [(2, 16, 16, 2), (3, 8, 12, 4), (4, 32, 32, 4), (2, 64, 64, 4)])
def test_conv2d_winograd_fwd_dgrad_wgrad(N, H, W, m):
    """The same Winograd path with one depth tap: Conv2d 128->128 k3 p1 (enc_b.blocks.4, dec.blocks.0 :113,140)."""
    from faceoff_amd import ops
    rng = np.random.default_rng(700 + N + H + m)
    x = _rand(rng, N, 128, H, W).requires_grad_(True)
    w = _rand(rng, 128, 128, 3, 3, scale=0.03).requires_grad_(True)
    b = _rand(rng, 128, scale=0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1)
    gy = _rand(rng, *y.shape)
    y.backward(gy)
    dev = _dev()
    xg, gyg, wg, bg = _nhwc(x.detach()), _nhwc(gy), w.detach().to(dev), b.detach().to(dev)
    out = torch.empty_like(xg)
    V = ops.conv3d_winograd(xg, ops.wino_filter(wg, m=m), bg, out, T=1, cin=128, cout=128, m=m, kd=1, keep_v=True)
    tol = 2e-5 if m == 2 else 2e-4
    _close(out.permute(0, 3, 1, 2), y, rtol=tol, what="winograd conv2d fwd")
    gx = torch.empty_like(xg)
    ops.conv3d_winograd(gyg, ops.wino_filter(wg, dgrad=True, m=m), None, gx, T=1, cin=128, cout=128, m=m, kd=1)
    _close(gx.permute(0, 3, 1, 2), x.grad, rtol=tol, what="winograd conv2d dgrad")
    if ops.wino_wgrad_ok(H, W, N, 1, m, kd=1):
        dw, db = torch.empty_like(wg), torch.empty_like(bg)
        ops.conv3d_wgrad_winograd(gyg, xg, dw, db, T=1, a_real=128, b_real=128, V=V, m=m, kd=1)
        _close(dw, w.grad, rtol=tol, what="winograd conv2d wgrad")
        _close(db, b.grad, rtol=2e-5, what="winograd conv2d bias grad")


def test_vq_assign_teacher_forced_indices_and_reproducible_commitment_sum():
    """fo_vq_assign2(..., forced_ind, ...): the search is skipped and the GIVEN codes are used -- index output = the forced codes, straight-through
    value x + (embed[:, ind] - x), commitment sum and EMA statistics of THAT assignment (Quantize.forward :55-78 with :54 replaced), in the fp32
    and the fp32 + bf16-copy forms; a code that is not the nearest one is taken as given.  The commitment sum is an ordered sum of
    per-workgroup partials: the same bits on every launch, and overwritten (a stale value in stats[0] does not leak in)."""
    from faceoff_amd import ops
    dev = _dev()
    rng = np.random.default_rng(21)
    x_np = rng.standard_normal((5, 9, 7, 64)).astype(np.float32) * 0.8
    e_np = rng.standard_normal((64, 512)).astype(np.float32)
    forced_np = rng.integers(0, 512, size=(5, 9, 7)).astype(np.int64)
    x, embed, forced = torch.from_numpy(x_np).to(dev), torch.from_numpy(e_np).to(dev), torch.from_numpy(forced_np).to(dev)
    embedT, enorm = ops.vq_prepare(embed)
    qe = torch.from_numpy(e_np.T[forced_np])                                   # [.., 64] = embed[:, ind]
    want_q = (x_np + (qe.numpy() - x_np)).astype(np.float32)
    want_sq = float(((qe.numpy().astype(np.float64) - x_np) ** 2).sum())
    runs = []
    for rep in range(3):
        q = torch.empty_like(x)
        stats = torch.full((1 + 512 + 512 * 64,), 123.0, device=dev)           # stale values everywhere
        ind = ops.vq_assign(x, embedT, enorm, q, stats, True, force_ind=forced)
        assert torch.equal(ind, forced)
        assert np.array_equal(q.cpu().numpy(), want_q)
        np.testing.assert_allclose(stats[0].item(), want_sq, rtol=1e-5)
        counts = np.bincount(forced_np.ravel(), minlength=512).astype(np.float32)
        assert np.array_equal(stats[1:513].cpu().numpy(), counts)
        runs.append(stats[0].item())
    assert runs[0] == runs[1] == runs[2]
    q32, q16 = torch.empty_like(x), torch.empty(x.shape, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(1 + 512 + 512 * 64, device=dev)
    ind = ops.vq_assign_bf16out(x, embedT, enorm, q32, q16, stats, False, force_ind=forced)
    assert torch.equal(ind, forced) and np.array_equal(q32.cpu().numpy(), want_q) and torch.equal(q16, q32.bfloat16()) and stats[0].item() == runs[0]
    free = ops.vq_assign(x, embedT, enorm, torch.empty_like(x), torch.zeros_like(stats), False)
    assert (free != forced).float().mean().item() > 0.9                        # (the forced codes really were not the nearest ones)
    # out-of-range codes (a -1 sentinel, c + 512): the kernel masks the index into [0, 512) -- memory-safe, but a silently aliased code -- so the binding
    # range-checks every DISTINCT forced tensor once (storage + version; one host sync the first time it is seen, none afterwards: ADVICE r05)
    for bad in (forced + 512, torch.where(forced == forced.flatten()[0], torch.full_like(forced, -1), forced)):
        with pytest.raises(ValueError):
            ops.vq_assign(x, embedT, enorm, torch.empty_like(x), stats, False, force_ind=bad)
    with pytest.raises(ValueError):                     # (the shape check is free and always on)
        ops.vq_assign(x, embedT, enorm, torch.empty_like(x), stats, False, force_ind=forced[:1])
    # a tensor that passed once is not read back again ... until it is modified in place (its version counter moves)
    seen = len(ops._forced_seen)
    ops.vq_assign(x, embedT, enorm, torch.empty_like(x), torch.zeros_like(stats), False, force_ind=forced)
    assert len(ops._forced_seen) == seen
    poisoned = forced.clone()
    ops.vq_assign(x, embedT, enorm, torch.empty_like(x), torch.zeros_like(stats), False, force_ind=poisoned)
    poisoned[0, 0, 0] = 700
    with pytest.raises(ValueError):
        ops.vq_assign(x, embedT, enorm, torch.empty_like(x), torch.zeros_like(stats), False, force_ind=poisoned)


def test_vq_assign_bit_exact_and_golden(golden_dir):
    """Indices bit-exact vs oracle/vq_oracle.c and vs the reference's own (golden) indices."""
    from faceoff_amd import ops
    from oracle import vq_c
    g = np.load(os.path.join(golden_dir, "quantize_kat.npz"))
    dev = _dev()
    for x_np, embed_np, tag in [(g["x"], g["embed"], "kat"),
                                (np.random.default_rng(9).standard_normal((3, 7, 5, 64)).astype(np.float32) * 0.7,
                                 np.random.default_rng(10).standard_normal((64, 512)).astype(np.float32), "ragged")]:
        want = vq_c.assign(x_np, embed_np)
        x = torch.from_numpy(x_np).to(dev)
        embed = torch.from_numpy(embed_np).to(dev)
        embedT, enorm = ops.vq_prepare(embed)
        q = torch.empty_like(x)
        stats = torch.zeros(1 + 512 + 512 * 64, device=dev)
        ind = ops.vq_assign(x, embedT, enorm, q, stats, True)
        assert np.array_equal(ind.cpu().numpy(), want["ind"]), tag
        assert np.array_equal(q.cpu().numpy(), want["q_ste"]), tag           # x + (q - x) bit-exact
        np.testing.assert_allclose(stats[0].item(), want["sq_sum"], rtol=1e-5)
        assert np.array_equal(stats[1:513].cpu().numpy(), want["counts"])
        np.testing.assert_allclose(stats[513:].view(512, 64).t().cpu().numpy(), want["esum"], rtol=1e-4, atol=1e-5)
        if tag == "kat":
            assert np.array_equal(ind.cpu().numpy().astype(np.int16), g["train_ind"])
            np.testing.assert_allclose(q.cpu().numpy(), g["train_quantize"], rtol=1e-6)
            # EMA update (:66-75) against the reference's post-step buffers
            cs = torch.from_numpy(g["cluster_size0"]).to(dev)
            ea = (embed * cs[None, :]).contiguous()
            emb = embed.clone()
            ops.vq_ema(emb, cs, ea, stats)
            np.testing.assert_allclose(cs.cpu().numpy(), g["train_cluster_size_after"], rtol=1e-6)
            np.testing.assert_allclose(ea.cpu().numpy(), g["train_embed_avg_after"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(emb.cpu().numpy(), g["train_embed_after"], rtol=1e-5, atol=1e-6)
            # backward: gx = g + gdiff * 2 (x - q)/numel
            gq = torch.from_numpy(g["train_gout"]).to(dev)
            gx = torch.empty_like(x)
            ops.vq_bwd(gq, x, q, torch.full((1,), 3.0, device=dev), gx)
            np.testing.assert_allclose(gx.cpu().numpy(), g["train_gx"], rtol=1e-5, atol=1e-7)
            # decode_code gather
            q2 = torch.empty_like(x)
            ops.vq_gather(ind, embedT, q2)
            np.testing.assert_allclose(q2.cpu().numpy(), embed_np.T[want["ind"]], rtol=0)


def test_vq_large_property():
    """BASELINE-size property check (655 360 vectors): every chosen code is a true nearest code in fp64
    up to fp32 rounding of the distance, counts sum to Nvec, esum sums to sum(x)."""
    from faceoff_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(3)
    nvec = 160 * 64 * 64
    x = (torch.randn(nvec, 64, generator=gen) * 0.5).to(dev).view(160, 64, 64, 64)
    embed = torch.randn(64, 512, generator=gen).to(dev) * 0.5
    embedT, enorm = ops.vq_prepare(embed)
    q = torch.empty_like(x)
    stats = torch.zeros(1 + 512 + 512 * 64, device=dev)
    ind = ops.vq_assign(x, embedT, enorm, q, stats, True)
    torch.cuda.synchronize()
    assert stats[1:513].sum().item() == nvec
    xf = x.view(-1, 64)
    np.testing.assert_allclose(stats[513:].view(512, 64).sum(0).cpu().numpy(), xf.sum(0).cpu().numpy(), rtol=2e-3, atol=0.5)
    sl = slice(0, 65536)
    xd, ed = xf[sl].double(), embed.double()
    dist = xd.pow(2).sum(1, keepdim=True) - 2 * xd @ ed + ed.pow(2).sum(0, keepdim=True)
    best = dist.min(1).values
    chosen = dist.gather(1, ind.view(-1)[sl].unsqueeze(1)).squeeze(1)
    assert (chosen - best).max().item() <= 2e-5 * dist.abs().max().item()
    assert torch.equal(q.view(-1, 64)[sl], xf[sl] + (embedT[ind.view(-1)[sl]] - xf[sl]))


def test_layout_mse_adam():
    from faceoff_amd import ops
    dev = _dev()
    rng = np.random.default_rng(4)
    x = _rand(rng, 3, 6, 10, 12)
    y = ops.nchw_to_nhwc(x.to(dev), cpad=8)
    assert torch.equal(y[..., :6].cpu(), x.permute(0, 2, 3, 1)) and y[..., 6:].abs().max().item() == 0
    assert torch.equal(ops.nhwc_to_nchw(y, 6).cpu(), x)
    gt = _rand(rng, 3, 3, 10, 12)
    acc = torch.zeros(1, device=dev)
    ops.mse_slice_fwd(y, gt.to(dev), acc)
    want = F.mse_loss(x[:, :3], gt)
    np.testing.assert_allclose(acc.item() / gt.numel(), want.item(), rtol=1e-5)
    gdec = torch.empty_like(y)
    ops.mse_slice_bwd(y, gt.to(dev), torch.full((1,), 0.5, device=dev), gdec)
    wantg = torch.zeros(3, 8, 10, 12)
    wantg[:, :3] = 0.5 * 2 * (x[:, :3] - gt) / gt.numel()
    np.testing.assert_allclose(gdec.permute(0, 3, 1, 2).cpu().numpy(), wantg.numpy(), rtol=1e-5, atol=1e-9)
    # Adam vs torch.optim.Adam, 3 steps
    p0 = _rand(rng, 1000)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=3e-4)
    p, m, v = p0.to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for step in range(1, 4):
        gcpu = _rand(rng, 1000)
        p_ref.grad = gcpu.clone()
        opt.step()
        ops.adam_flat(p, gcpu.to(dev), m, v, 3e-4, step)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_vector_alu_lane_moves_equal_the_shuffles():
    """common.h lane_xor<K> / group_sum_valu (DPP quad_perm / row_shl+row_shr / row_ror, v_permlane16/32_swap) against __shfl_xor (ds_bpermute) on
    one wave of distinct floats and ints, every distance 1..32 and the 8- and 64-lane butterfly sums: the reductions that were switched to them
    (LPIPS heads, discriminator head, VQ, weight-gradient column sums) keep their bits."""
    from faceoff_amd import _lib, ops
    bad = torch.full((64,), -1, dtype=torch.int32, device=_dev())
    _lib.call("fo_selftest_lane_moves", ops._ptr(bad), ops._stream())
    assert bad.cpu().tolist() == [0] * 64, bad.cpu().tolist()


@pytest.mark.parametrize("N,T,Ht,Wt,kd,ci,co", [(10, 5, 8, 8, 3, 128, 128), (4, 2, 16, 16, 3, 128, 128), (8, 4, 12, 12, 3, 64, 128),
                                                 (6, 1, 8, 8, 3, 128, 128), (8, 1, 16, 8, 1, 128, 128), (32, 4, 4, 4, 3, 96, 256)])
def test_wino_gemm_equals_the_banked_implicit_gemm(N, T, Ht, Wt, kd, ci, co):
    """fo_wino_gemm (persistent plane-stack GEMM: tiles that straddle frames and clips, skipped depth taps, rows per
    frame that are / are not powers of two, several tiles per workgroup) against fo_conv_igemm_banked on the same
    planes: the same k-ordered fp32 MFMA chain, so the results must be the same BITS."""
    import ctypes as C
    from faceoff_amd import ops, _lib
    planes = 16
    g = torch.Generator(device="cuda").manual_seed(5)
    V = torch.randn((planes * N * Ht * Wt, ci), device="cuda", generator=g)
    U = torch.randn((planes, co, kd, ci), device="cuda", generator=g) * 0.05
    M0 = torch.full((planes * N * Ht * Wt, co), 7.0, device="cuda")
    M1 = torch.full_like(M0, -3.0)
    d = ops._desc(N=planes * N, T=T if kd > 1 else 1, Hin=Ht, Win=Wt, Hm=Ht, Wm=Wt, Hout=Ht, Wout=Wt, Cin=ci, Cout=co, KD=kd, KH=1, KW=1,
                  stride=1, padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=ci, ldOut=co, ldMask=0, ldAdd=0, flags=0)
    _lib.call("fo_conv_igemm_banked", C.byref(d), ops._ptr(V), ops._ptr(U), ops._ptr(M0), N, ops._stream())
    _lib.call("fo_wino_gemm", ops._ptr(V), ops._ptr(U), ops._ptr(M1), planes, N, T if kd > 1 else 1, Ht * Wt, ci, co, kd, ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(M0, M1), (M0 - M1).abs().max().item()
    # and against plain torch on one plane (fp64 reference of the definition)
    p = 3
    v = V.view(planes, N // T, T, Ht * Wt, ci)[p].double()
    u = U[p].double()
    ref = torch.zeros((N // T, T, Ht * Wt, co), device="cuda", dtype=torch.float64)
    for k in range(kd):
        dt = k - kd // 2
        for t in range(T):
            if 0 <= t + dt < T:
                ref[:, t] += v[:, t + dt] @ u[:, k].T
    got = M1.view(planes, N // T, T, Ht * Wt, co)[p].double()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("N,H,W,out_relu", [(3, 16, 16, False), (2, 20, 12, True), (1, 7, 9, False)])
def test_fused_resblock_forward_equals_the_two_launch_form(N, H, W, out_relu):
    """fo_resblock_fwd (3x3 -> ReLU -> 1x1 -> + input in one launch, hidden tile through LDS) against torch-CPU ResBlock
    arithmetic (reference models/vqvae_conv3d_latent.py:86-101) and against the engine's two-launch form."""
    from faceoff_amd import ops
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn((N, 128, H, W), generator=g)
    w1, b1 = torch.randn((32, 128, 3, 3), generator=g) * 0.05, torch.randn(32, generator=g) * 0.1
    w3, b3 = torch.randn((128, 32, 1, 1), generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
    h_ref = torch.relu(torch.nn.functional.conv2d(torch.relu(x), w1, b1, padding=1))
    o_ref = torch.nn.functional.conv2d(h_ref, w3, b3) + x
    if out_relu:
        o_ref = torch.relu(o_ref)
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp1, wp3 = ops.pack_conv(w1.cuda()), ops.pack_conv(w3.cuda())
    hb = torch.full((N, H, W, 32), 5.0, device="cuda")
    out = torch.full((N, H, W, 128), -7.0, device="cuda")
    ops.resblock_fwd(xc, wp1, b1.cuda(), wp3, b3.cuda(), hb, out, out_relu)
    assert (hb.permute(0, 3, 1, 2).cpu() - h_ref).abs().max().item() <= 2e-5 * h_ref.abs().max().item()
    assert (out.permute(0, 3, 1, 2).cpu() - o_ref).abs().max().item() <= 2e-5 * o_ref.abs().max().item()
    hb2, out2 = torch.empty_like(hb), torch.empty_like(out)
    ops.conv_igemm(xc, wp1, b1.cuda(), hb2, cin=128, cout=32, flags=ops.FO_IN_RELU | ops.FO_OUT_RELU)
    ops.conv_igemm(hb2, wp3, b3.cuda(), out2, k=(1, 1, 1), pad=(0, 0, 0), cin=32, cout=128, flags=ops.FO_OUT_RELU if out_relu else 0, add=xc)
    assert torch.equal(hb, hb2)                                   # same k-ordered MFMA chain for the 3x3 half
    assert (out - out2).abs().max().item() <= 1e-6 * out2.abs().max().item()


@pytest.mark.parametrize("N,H,W,out_relu", [(3, 16, 32, False), (2, 6, 64, True), (1, 32, 32, False)])
def test_resblock_halo_tile_kernel_vs_torch_and_vs_the_tiled_form(N, H, W, out_relu, monkeypatch):
    """resblock_halo_fwd_kernel (csrc/resblock_halo.hip: input patch staged once per 2 x 32-pixel tile, contraction split over the waves with each
    wave's filter slice resident in registers; taken at C2 sizes, forced here) against torch-CPU ResBlock arithmetic (reference
    models/vqvae_conv3d_latent.py:86-101) and against the tiled fo_resblock_fwd: same bound, results equal up to summation order."""
    import subprocess, sys, os, json
    code = r"""
import sys, json, torch
sys.path.insert(0, %r)
from faceoff_amd import ops
N, H, W, out_relu = %d, %d, %d, %s
g = torch.Generator().manual_seed(N * 100 + H)
x = torch.randn((N, 128, H, W), generator=g)
w1, b1 = torch.randn((32, 128, 3, 3), generator=g) * 0.05, torch.randn(32, generator=g) * 0.1
w3, b3 = torch.randn((128, 32, 1, 1), generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
h_ref = torch.relu(torch.nn.functional.conv2d(torch.relu(x), w1, b1, padding=1))
o_ref = torch.nn.functional.conv2d(h_ref, w3, b3) + x
if out_relu: o_ref = torch.relu(o_ref)
xc = x.permute(0, 2, 3, 1).contiguous().cuda()
wide = torch.zeros((N, H, W, 192), device="cuda"); wide[..., 64:192] = xc          # also through a channel-slice view (ld = 192)
wp1, wp3 = ops.pack_conv(w1.cuda()), ops.pack_conv(w3.cuda())
res = {}
for name, xin in (("dense", xc), ("slice", wide[..., 64:192])):
    hb = torch.full((N, H, W, 32), 5.0, device="cuda"); out = torch.full((N, H, W, 128), -7.0, device="cuda")
    ops.resblock_fwd(xin, wp1, b1.cuda(), wp3, b3.cuda(), hb, out, out_relu)
    res[name] = ((hb.permute(0, 3, 1, 2).cpu() - h_ref).abs().max().item() / h_ref.abs().max().item(),
                 (out.permute(0, 3, 1, 2).cpu() - o_ref).abs().max().item() / o_ref.abs().max().item())
# the block's first conv backwards (conv3x3_c32_halo_kernel, csrc/resblock_bwd.hip): g_x = conv3x3^T(g_h) * (x > 0) + g_out
gh = torch.randn((N, 32, H, W), generator=g); go = torch.randn((N, 128, H, W), generator=g)
gx_ref = (torch.nn.functional.conv_transpose2d(gh.double(), w1.double(), padding=1) * (x > 0) + go.double())
wpd = ops.pack_conv_dgrad(w1.cuda().reshape(32, 128, -1))
ghc, goc = gh.permute(0, 2, 3, 1).contiguous().cuda(), go.permute(0, 2, 3, 1).contiguous().cuda()
wide2 = torch.zeros((N, H, W, 64), device="cuda"); wide2[..., 16:48] = ghc
for name, gin, xin in (("dgrad dense", ghc, xc), ("dgrad slice", wide2[..., 16:48], wide[..., 64:192])):
    gx = torch.full((N, H, W, 128), 11.0, device="cuda")
    ops.conv_igemm(gin, wpd, None, gx, k=(1, 3, 3), stride=1, pad=(0, 1, 1), cin=32, cout=128, mask=xin, add=goc)
    e = (gx.permute(0, 3, 1, 2).cpu().double() - gx_ref).abs().max().item() / gx_ref.abs().max().item()
    res[name] = (e, e)
# ... and its filter / bias gradient (resblock_wgrad1_halo_kernel): dW1 = sum relu(x)[pixel + tap] (x) g_h[pixel], db1 = sum g_h
xr = torch.relu(x).double()
dw_ref = torch.nn.grad.conv2d_weight(xr, (32, 128, 3, 3), gh.double(), padding=1)
db_ref = gh.double().sum((0, 2, 3))
bound = torch.nn.grad.conv2d_weight(xr.abs(), (32, 128, 3, 3), gh.double().abs(), padding=1).max().item()
for name, gin, xin in (("wgrad dense", ghc, xc), ("wgrad slice", wide2[..., 16:48], wide[..., 64:192])):
    dw = torch.full((32, 128, 3, 3), 7.0, device="cuda"); db = torch.full((32,), 7.0, device="cuda")
    ops.conv_wgrad(gin, xin, dw, db, k=(1, 3, 3), stride=1, pad=(0, 1, 1), a_real=32, b_real=128, in_relu=True)
    res[name] = ((dw.cpu().double() - dw_ref).abs().max().item() / bound, (db.cpu().double() - db_ref).abs().max().item() / gh.abs().sum((0, 2, 3)).max().item())
print(json.dumps(res))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), N, H, W, out_relu)
    outs = {}
    for mode in ("halo", "tiled"):       # (the switch is read once per process)
        env = dict(os.environ, **({"FACEOFF_FORCE_RESBLOCK_HALO": "1"} if mode == "halo" else {"FACEOFF_NO_RESBLOCK_HALO": "1"}))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for mode, res in outs.items():
        for name, (eh, eo) in res.items():
            assert eh <= 2e-5 and eo <= 2e-5, (mode, name, eh, eo)


@pytest.mark.parametrize("N,H,W", [(2, 8, 8), (3, 7, 9), (160, 16, 16)])
def test_resblock_bwd_conv3_one_pass_vs_torch_and_vs_the_three_launches(N, H, W):
    """fo_resblock_bwd_conv3 (csrc/resblock_bwd.hip): the data, filter and bias gradients of a ResBlock's 1x1 convolution (reference
    models/vqvae_conv3d_latent.py:94-95) in one pass over the output gradient -- against torch-CPU in float64 and against the three separate
    launches it replaces; ragged pixel counts (not a multiple of the 64-pixel tile); through a channel-slice view; twice = bit-identical."""
    from faceoff_amd import ops
    gen = torch.Generator().manual_seed(N * 10 + W)
    g = torch.randn((N, H, W, 128), generator=gen)
    h = torch.relu(torch.randn((N, H, W, 32), generator=gen))          # post-ReLU hidden activation: about half zeros
    w3 = torch.randn((128, 32, 1, 1), generator=gen) * 0.1
    W2 = w3.reshape(128, 32).double()
    G, Hh = g.reshape(-1, 128).double(), h.reshape(-1, 32).double()
    gh_ref = (G @ W2) * (Hh > 0)
    dw_ref, db_ref = G.t() @ Hh, G.sum(0)
    gc, hc = g.cuda(), h.cuda()
    wide = torch.zeros((N, H, W, 192), device="cuda"); wide[..., 32:160] = gc
    wp3 = ops.pack_conv(w3.cuda())
    outs = []
    for gin in (gc, wide[..., 32:160], gc):
        gh = torch.full((N, H, W, 32), 3.0, device="cuda"); dw = torch.full((128, 32), 9.0, device="cuda"); db = torch.full((128,), 9.0, device="cuda")
        ops.resblock_bwd_conv3(gin, hc, wp3, gh, dw, db)
        outs.append((gh.cpu(), dw.cpu(), db.cpu()))
        scale = lambda t: t.abs().max().item()
        assert (gh.cpu().reshape(-1, 32).double() - gh_ref).abs().max().item() <= 2e-6 * scale(gh_ref)
        assert (dw.cpu().double() - dw_ref).abs().max().item() <= 2e-6 * max(scale(dw_ref), (G.abs().t() @ Hh).max().item())
        assert (db.cpu().double() - db_ref).abs().max().item() <= 2e-6 * G.abs().sum(0).max().item()
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)                                   # fixed tile walk, fixed slab order
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                   # the pixel stride does not enter the arithmetic
    # the launches it replaces
    gh2 = torch.empty((N, H, W, 32), device="cuda"); dw2 = torch.empty((128, 32, 1, 1), device="cuda"); db2 = torch.empty((128,), device="cuda")
    wpd = ops.pack_conv_dgrad(w3.cuda().reshape(128, 32, -1))
    ops.conv_igemm(gc, wpd, None, gh2, k=(1, 1, 1), stride=1, pad=(0, 0, 0), cin=128, cout=32, mask=hc)
    ops.conv_wgrad(gc, hc, dw2, db2, k=(1, 1, 1), stride=1, pad=(0, 0, 0), a_real=128, b_real=32)
    assert (outs[0][0] - gh2.cpu()).abs().max().item() <= 2e-6 * gh2.abs().max().item()
    assert (outs[0][1] - dw2.cpu().reshape(128, 32)).abs().max().item() <= 4e-6 * (G.abs().t() @ Hh).max().item()
    assert (outs[0][2] - db2.cpu()).abs().max().item() <= 4e-6 * G.abs().sum(0).max().item()
