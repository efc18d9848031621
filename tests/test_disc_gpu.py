"""MoCoGAN-HD discriminators on the MI355X (BASELINE config 5 building blocks) against (1) golden outputs of the reference's
own ModelD_3d / ModelD_img / Relativistic_Average_LSGAN (tests/golden/disc_kat.npz) and (2) the CPU oracle: multiscale patch
logits, the discriminator- and generator-form losses, gradients of all 20 parameter tensors and of the fake input,
InstanceNorm running statistics, the Adam(0.5, 0.999) update; plus per-kernel checks of the general N-d conv kernels."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_disc_state

pytestmark = pytest.mark.gpu
SUB = 211


def _sub(t):
    return t.detach().reshape(-1)[::SUB].cpu().numpy()


def _cl(x):
    """[N,C,(D,)H,W] torch tensor -> channels-last [N,D,H,W,32] on the GPU, zero-padded channels."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    N, Cc, D, H, W = x.shape
    out = torch.zeros((N, D, H, W, 32), device="cuda")
    out[..., :Cc] = x.permute(0, 2, 3, 4, 1).cuda()
    return out.contiguous()


def _inputs(g, tag, dims):
    F, H, W = (int(g[f"{tag}_{k}"]) for k in "FHW")
    rng = np.random.default_rng(int(g[f"{tag}_seed_x"]))
    shape = (1, 6, F - 1, H, W) if dims == 3 else (1, 6, H, W)
    real = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    return real, fake, F


@pytest.mark.parametrize("tag,dims", [("v", 3), ("i", 2)])
def test_discriminator_engine_vs_reference_golden(golden_dir, tag, dims):
    from faceoff_amd.disc import DiscEngine, ralsgan_pair
    g = np.load(os.path.join(golden_dir, "disc_kat.npz"))
    real, fake, F = _inputs(g, tag, dims)
    sd = make_disc_state(int(g[f"{tag}_seed_w"]), dims)
    eng = DiscEngine(sd, "cuda:0", dims=dims, n_frames=F - 1)
    x = torch.cat([_cl(fake), _cl(real)], 0)                       # sample 0 = fake, 1 = real (module calls: fake, then real)
    S = eng.forward(x, training=True, sample_order=[0, 1])
    for s in range(2):
        lg = S["logits"][s][..., 0].cpu().numpy()
        for n, key in ((0, f"{tag}_fake_s{s}_logits"), (1, f"{tag}_real_s{s}_logits")):
            want = g[key].reshape(lg[n].shape)
            assert np.abs(lg[n] - want).max() <= 1e-3 * np.abs(want).max(), key
    # discriminator-form loss: 0.5 * (crit(real, fake, True) + crit(fake, real, False)), both samples carry gradient
    loss = torch.zeros(1, device="cuda")
    g_logits = ralsgan_pair(S["logits"], 1, 0, 1.0, 0.0, 0.5, loss)
    np.testing.assert_allclose(loss.item(), float(g[f"{tag}_d_loss"]), rtol=1e-3)
    gx = eng.backward(S, g_logits, param_grads=True, input_grad=True)
    names = [str(n) for n in g[f"{tag}_param_names"]]
    assert names == list(eng.grads)
    got = np.concatenate([_sub(eng.grads[n]) for n in names])
    want = g[f"{tag}_d_grad_sub"]
    off, worst = 0, (0.0, "")
    for i, n in enumerate(names):                                   # every tensor on its own scale (rms | max of the subsample)
        k = len(_sub(eng.grads[n]))
        w_, g_ = want[off:off + k], got[off:off + k]
        off += k
        scale = max(np.sqrt(g[f"{tag}_d_grad_stats"][i, 1] / eng.grads[n].numel()), np.abs(w_).max())
        if scale < 1e-6 * np.abs(want).max():                       # bias in front of an InstanceNorm: zero up to rounding
            assert np.abs(g_).max() <= 1e-5 * np.abs(want).max(), n
            continue
        err = np.abs(g_ - w_).max() / scale
        worst = max(worst, (float(err), n))
        assert err <= 1e-3, (n, err)
    gfake = gx[0, ..., :6].permute(3, 0, 1, 2).reshape(fake.shape)
    wsub = g[f"{tag}_d_gfake_sub"]
    assert np.abs(_sub(gfake) - wsub).max() <= 1e-3 * np.abs(wsub).max()
    assert torch.equal(gx[..., 6:], torch.zeros_like(gx[..., 6:]))
    for k, b in eng.state_dict().items():                           # running statistics after (fake, real)
        if "running" in k:
            np.testing.assert_allclose(b.cpu().numpy(), g[f"{tag}_buf.{k}"], rtol=1e-3, atol=1e-6)
    # Adam(lr=1e-4, betas=(0.5, 0.999)) on those gradients
    eng.adam_step(1e-4)
    after = np.concatenate([_sub(eng.params[n]) for n in names])
    big = np.abs(want) > 1e-3 * np.abs(want).max()
    assert np.abs(after - g[f"{tag}_param_after_sub"])[big].max() <= 5e-6
    # generator-form loss: 0.5 * (crit(fake, real, True) + crit(real, fake, False)); gradient to the fake input only
    eng2 = DiscEngine(sd, "cuda:0", dims=dims, n_frames=F - 1)
    S2 = eng2.forward(x, training=True, sample_order=[0, 1])
    loss2 = torch.zeros(1, device="cuda")
    g2 = ralsgan_pair(S2["logits"], 0, 1, 1.0, 0.0, 0.5, loss2, want_gb=False)
    np.testing.assert_allclose(loss2.item(), float(g[f"{tag}_g_loss"]), rtol=1e-3)
    gx2 = eng2.backward(S2, g2, param_grads=False, input_grad=True)
    gf2 = gx2[0, ..., :6].permute(3, 0, 1, 2).reshape(fake.shape)
    wsub2 = g[f"{tag}_g_gfake_sub"]
    assert np.abs(_sub(gf2) - wsub2).max() <= 1e-3 * np.abs(wsub2).max()
    print(f"[disc {tag}] d_loss {loss.item():.6f} (reference {float(g[f'{tag}_d_loss']):.6f}); worst parameter-gradient rel err {worst}")


@pytest.mark.parametrize("dims,N,Cin,Cout,size,stride", [(3, 2, 32, 64, (5, 9, 11), 2), (3, 1, 64, 128, (4, 10, 7), 1), (2, 2, 96, 64, (1, 13, 12), 2),
                                                         (3, 1, 64, 1, (3, 6, 6), 1), (2, 1, 32, 200, (1, 17, 9), 1),
                                                         # enough rows for the filter gradient to run as row slices + the reduce launch
                                                         (2, 2, 32, 64, (1, 40, 44), 1), (3, 1, 32, 64, (3, 20, 20), 1), (3, 2, 32, 64, (8, 21, 23), 2)])
@pytest.mark.parametrize("ksize,padding", [(4, 2), (3, 1), (5, 2), (2, 0), (3, 3)])
def test_convnd_forward_transposed_and_wgrad_vs_torch(dims, N, Cin, Cout, size, stride, ksize, padding):
    """fo_convnd forward / transposed (gather data gradient, phase-major rows) / fo_wgradnd against torch-CPU convolutions: the discriminators' k4 p2
    (mocoganhd_video_disc.py:133-158) and other kernel sizes / paddings (the walks leave out the taps and frames that only see padding: generic code)."""
    if (ksize, padding) != (4, 2) and (Cout in (1, 200) or size[1] >= 40):
        pytest.skip("the other kernel sizes run on three of the shapes")
    from faceoff_amd import _lib, ops
    from faceoff_amd._lib import ConvNdDesc, FO_BIAS
    g = torch.Generator().manual_seed(dims * 100 + Cin + Cout)
    D, H, W = size
    x = torch.randn((N, Cin, D, H, W), generator=g)
    K, P = ksize, padding
    kshape = (K, K, K) if dims == 3 else (1, K, K)
    w = torch.randn((Cout, Cin) + kshape, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    pad = (P, P, P) if dims == 3 else (0, P, P)
    if any((n + 2 * p - k) < 0 for n, p, k in zip(size, pad, kshape)):
        pytest.skip("kernel larger than the padded input")
    st = (stride,) * 3 if dims == 3 else (1, stride, stride)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(xr, wr, b, stride=st, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    Do, Ho, Wo = y.shape[2:]
    ldo = max(32, (Cout + 31) // 32 * 32)
    d = ConvNdDesc(N=N, Ds=D, Hs=H, Ws=W, Cs=Cin, ldS=Cin, Dd=Do, Hd=Ho, Wd=Wo, Cd=Cout, ldD=ldo, KD=kshape[0], KH=K, KW=K,
                   sD=st[0], sH=st[1], sW=st[2], pD=pad[0], pH=P, pW=P, ldMask=0, flags=FO_BIAS, slope=0.2)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    taps = kshape[0] * K * K
    wc = w.reshape(Cout, Cin, taps).contiguous().cuda()
    wp = torch.empty(((Cout + 63) // 64 * 64) * taps * Cin, device="cuda")
    _lib.call("fo_pack_convnd", ops._ptr(wc), ops._ptr(wp), Cout, Cin, taps, 0, ops._stream())
    out = torch.zeros((N, Do, Ho, Wo, ldo), device="cuda")
    _lib.call("fo_convnd", C.byref(d), 0, ops._ptr(xc), ops._ptr(wp), ops._ptr(b.cuda()), None, ops._ptr(out), None, C.c_int64(0), ops._stream())
    got = out[..., :Cout].permute(0, 4, 1, 2, 3).cpu()
    assert (got - y.detach()).abs().max().item() <= 2e-5 * y.detach().abs().max().item()
    # the same launch with its contraction sliced over workgroups (FO_KSPLIT; through a workspace, epilogue in the reduce launch):
    # equal up to summation order, and bit-reproducible
    from faceoff_amd._lib import FO_KSPLIT, FO_OUT_LRELU
    d.flags = FO_BIAS | FO_OUT_LRELU | FO_KSPLIT
    nb = _lib.load().fo_convnd_ws_bytes(C.byref(d), 0)
    assert nb >= 0
    if nb:
        ws = torch.empty(nb // 4, device="cuda")
        outs = []
        for _ in range(2):
            o = torch.full((N, Do, Ho, Wo, ldo), 3.0, device="cuda")
            _lib.call("fo_convnd", C.byref(d), 0, ops._ptr(xc), ops._ptr(wp), ops._ptr(b.cuda()), None, ops._ptr(o), ops._ptr(ws), C.c_int64(nb), ops._stream())
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
        want = torch.nn.functional.leaky_relu(y.detach(), 0.2)
        assert (outs[0][..., :Cout].permute(0, 4, 1, 2, 3).cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        with pytest.raises(_lib.FaceoffHipError):          # sliced launch without its workspace
            _lib.call("fo_convnd", C.byref(d), 0, ops._ptr(xc), ops._ptr(wp), ops._ptr(b.cuda()), None, ops._ptr(o), None, C.c_int64(0), ops._stream())
    # transposed: source = gy padded to a multiple of 32 channels
    cs = (Cout + 31) // 32 * 32
    gc = torch.zeros((N, Do, Ho, Wo, cs), device="cuda")
    gc[..., :Cout] = gy.permute(0, 2, 3, 4, 1).cuda()
    wpt = torch.empty(((Cin + 63) // 64 * 64) * taps * cs, device="cuda")
    _lib.call("fo_pack_convnd", ops._ptr(wc), ops._ptr(wpt), Cout, Cin, taps, 1, ops._stream())
    dt = ConvNdDesc(N=N, Ds=Do, Hs=Ho, Ws=Wo, Cs=cs, ldS=cs, Dd=D, Hd=H, Wd=W, Cd=Cin, ldD=Cin, KD=kshape[0], KH=K, KW=K,
                    sD=st[0], sH=st[1], sW=st[2], pD=pad[0], pH=P, pW=P, ldMask=0, flags=0, slope=0.2)
    gin = torch.full((N, D, H, W, Cin), 9.0, device="cuda")
    _lib.call("fo_convnd", C.byref(dt), 1, ops._ptr(gc), ops._ptr(wpt), None, None, ops._ptr(gin), None, C.c_int64(0), ops._stream())
    got = gin.permute(0, 4, 1, 2, 3).cpu()
    assert (got - xr.grad).abs().max().item() <= 2e-5 * xr.grad.abs().max().item()
    # filter gradient
    d.flags = 0
    d.ldD = cs
    dw = torch.full((Cout, Cin, taps), 7.0, device="cuda")          # (overwritten, not accumulated into)
    ws = ops._workspace(_lib.load().fo_wgradnd_ws_bytes(C.byref(d)), dw.device)
    args = (C.byref(d), ops._ptr(gc), ops._ptr(xc), ops._ptr(dw), Cin, ops._ptr(ws), C.c_int64(ws.numel() * 4), ops._stream())
    _lib.call("fo_wgradnd", *args)
    want = wr.grad.reshape(Cout, Cin, taps)
    assert (dw.cpu() - want).abs().max().item() <= 5e-5 * want.abs().max().item()
    first = dw.clone()
    _lib.call("fo_wgradnd", *args)                                   # row slices are added in slice order: bit-reproducible
    assert torch.equal(first, dw)


@pytest.mark.parametrize("dims,N,Cin,size", [(3, 2, 512, (3, 9, 11)), (2, 2, 512, (1, 13, 12)), (3, 1, 256, (4, 6, 7)), (2, 3, 256, (1, 5, 18))])
def test_disc_head_kernels_vs_torch(dims, N, Cin, size):
    """fo_disc_head_{fwd,dgrad,wgrad} (csrc/disc_head.hip: the 512 -> 1 patch head, reference mocoganhd_video_disc.py:150-158, as dot products)
    against torch-CPU conv k4 s1 p2 in float64; the pixel's other 31 floats stay untouched; every result bit-identical between two calls."""
    from faceoff_amd import _lib, ops
    from faceoff_amd._lib import ConvNdDesc, FO_BIAS
    g = torch.Generator().manual_seed(dims * 1000 + Cin + N)
    D, H, W = size
    x = torch.randn((N, Cin, D, H, W), generator=g)
    kshape = (4, 4, 4) if dims == 3 else (1, 4, 4)
    w = torch.randn((1, Cin) + kshape, generator=g) * 0.05
    b = torch.randn(1, generator=g)
    pad = (2, 2, 2) if dims == 3 else (0, 2, 2)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = torch.nn.functional.conv3d(xr, wr, b.double(), stride=1, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    Do, Ho, Wo = y.shape[2:]
    taps = kshape[0] * 16
    d = ConvNdDesc(N=N, Ds=D, Hs=H, Ws=W, Cs=Cin, ldS=Cin, Dd=Do, Hd=Ho, Wd=Wo, Cd=1, ldD=32, KD=kshape[0], KH=4, KW=4, sD=1, sH=1, sW=1,
                   pD=pad[0], pH=2, pW=2, ldMask=0, flags=FO_BIAS, slope=0.2)
    xc = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    wc = w.reshape(1, Cin, taps).contiguous().cuda()
    wp = torch.empty(64 * taps * Cin, device="cuda")
    _lib.call("fo_pack_convnd", ops._ptr(wc), ops._ptr(wp), 1, Cin, taps, 0, ops._stream())
    gc = torch.zeros((N, Do, Ho, Wo, 32), device="cuda")
    gc[..., 0] = gy[:, 0].cuda()
    res = []
    for _ in range(2):
        out = torch.full((N, Do, Ho, Wo, 32), 7.0, device="cuda")
        _lib.call("fo_disc_head_fwd", C.byref(d), ops._ptr(xc), ops._ptr(wp), ops._ptr(b.cuda()), ops._ptr(out), ops._stream())
        gin = torch.full((N, D, H, W, Cin), 9.0, device="cuda")
        _lib.call("fo_disc_head_dgrad", C.byref(d), ops._ptr(gc), ops._ptr(wp), ops._ptr(gin), ops._stream())
        dw = torch.full((1, Cin, taps), 5.0, device="cuda")
        ws = torch.empty(_lib.load().fo_disc_head_wgrad_ws_bytes(C.byref(d)) // 4, device="cuda")
        _lib.call("fo_disc_head_wgrad", C.byref(d), ops._ptr(gc), ops._ptr(xc), ops._ptr(dw), Cin, ops._ptr(ws), C.c_int64(ws.numel() * 4), ops._stream())
        res.append((out.cpu(), gin.cpu(), dw.cpu()))
    out, gin, dw = res[0]
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    assert (out[..., 1:] == 7.0).all()
    yd = y.detach()[:, 0]
    assert (out[..., 0].double() - yd).abs().max().item() <= 2e-6 * yd.abs().max().item()
    assert (gin.permute(0, 4, 1, 2, 3).double() - xr.grad).abs().max().item() <= 2e-6 * xr.grad.abs().max().item()
    want = wr.grad.reshape(1, Cin, taps)
    bound = torch.nn.functional.conv3d(x.double().abs().transpose(0, 1), gy.double().abs().transpose(0, 1), padding=pad).max().item()
    assert (dw.double() - want).abs().max().item() <= 2e-6 * max(bound, want.abs().max().item())


def test_avgpool_instnorm_pairs_ralsgan_vs_torch():
    from faceoff_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    # AvgPool3d(3, stride (1,2,2), padding 1, count_include_pad=False) forward / backward, odd sizes
    x = torch.randn((1, 32, 5, 9, 12), generator=g, requires_grad=True)
    y = torch.nn.functional.avg_pool3d(x, 3, stride=(1, 2, 2), padding=1, count_include_pad=False)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xc = x.detach().permute(0, 2, 3, 4, 1)[0].contiguous().cuda()
    out = torch.empty(tuple(y.shape[2:]) + (32,), device="cuda")
    _lib.call("fo_avgpool3_fwd", ops._ptr(xc), ops._ptr(out), 5, 9, 12, 32, 32, 3, 1, 2, 2, ops._stream())
    assert (out.permute(3, 0, 1, 2).cpu() - y.detach()[0]).abs().max().item() <= 1e-6
    gx = torch.zeros_like(xc)
    _lib.call("fo_avgpool3_bwd", ops._ptr(gy[0].permute(1, 2, 3, 0).contiguous().cuda()), ops._ptr(gx), 5, 9, 12, 32, 32, 3, 1, 2, 2, ops._stream())
    assert (gx.permute(3, 0, 1, 2).cpu() - x.grad[0]).abs().max().item() <= 1e-6
    # InstanceNorm + LeakyReLU forward / backward + running statistics
    rows, Cc = 7 * 5 * 3, 48
    x = (torch.randn((1, Cc, 7, 5, 3), generator=g) * 2 + 0.7).requires_grad_(True)
    inorm = torch.nn.InstanceNorm3d(Cc, affine=False, track_running_stats=True)
    yy = torch.nn.functional.leaky_relu(inorm(x), 0.2)
    gyy = torch.randn(yy.shape, generator=g)
    yy.backward(gyy)
    xc = x.detach().permute(0, 2, 3, 4, 1).reshape(rows, Cc).contiguous().cuda()
    yc, st = torch.empty_like(xc), torch.empty(2 * Cc, device="cuda")
    run = torch.cat([torch.zeros(Cc), torch.ones(Cc)]).cuda()
    _lib.call("fo_instnorm_lrelu_fwd", ops._ptr(xc), Cc, ops._ptr(yc), Cc, C.c_int64(rows), Cc, C.c_float(1e-5), C.c_float(0.2), ops._ptr(st),
              ops._ptr(run), C.c_float(0.1), 0, ops._stream())
    want = yy.detach().permute(0, 2, 3, 4, 1).reshape(rows, Cc)
    assert (yc.cpu() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    np.testing.assert_allclose(run[:Cc].cpu().numpy(), inorm.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(run[Cc:].cpu().numpy(), inorm.running_var.numpy(), rtol=1e-5)
    gxc = torch.empty_like(xc)
    _lib.call("fo_instnorm_lrelu_bwd", ops._ptr(gyy.permute(0, 2, 3, 4, 1).reshape(rows, Cc).contiguous().cuda()), Cc, ops._ptr(yc), Cc, ops._ptr(st),
              ops._ptr(gxc), Cc, C.c_int64(rows), Cc, C.c_float(0.2), ops._stream())
    wantg = x.grad.permute(0, 2, 3, 4, 1).reshape(rows, Cc)
    assert (gxc.cpu() - wantg).abs().max().item() <= 2e-5 * wantg.abs().max().item()
    # frame pairing (both source layouts, flipped order) and its gradient
    from faceoff_amd.disc import make_pairs, pairs_backward
    F, H, W = 6, 4, 5
    fr = torch.randn((F, 3, H, W), generator=g)
    want = torch.cat((fr[0:1].expand(F - 1, 3, H, W), fr[1:]), dim=1)            # [F-1, 6, H, W]
    out = torch.empty((F - 1, H, W, 32), device="cuda")
    make_pairs(fr.cuda(), True, 0, 1, 1, F - 1, out)
    assert torch.equal(out[..., :6].permute(0, 3, 1, 2).cpu(), want) and torch.equal(out[..., 6:], torch.zeros_like(out[..., 6:]))
    nh = torch.zeros((F, H, W, 8), device="cuda")
    nh[..., :3] = fr.permute(0, 2, 3, 1).cuda()
    make_pairs(nh, False, 0, F - 1, -1, F - 1, out)                              # flip_video: reversed pair order
    assert torch.equal(out[..., :6].permute(0, 3, 1, 2).cpu(), torch.flip(want, [0]))
    gp = torch.randn((F - 1, H, W, 32), generator=g).cuda()
    gfr = torch.zeros((F, H, W, 8), device="cuda")
    pairs_backward(gp, 0, F - 1, -1, F - 1, gfr, scale=2.0)
    wg = torch.zeros((F, 3, H, W))
    gpp = torch.flip(gp[..., :6].permute(0, 3, 1, 2).cpu(), [0])                 # back to natural pair order
    wg[0] = gpp[:, :3].sum(0)
    wg[1:] += gpp[:, 3:]
    assert (gfr[..., :3].permute(0, 3, 1, 2).cpu() - 2.0 * wg).abs().max().item() <= 1e-5
    # relativistic average LSGAN, both ways, with gradients
    a = torch.randn(37, generator=g, requires_grad=True)
    b = torch.randn(37, generator=g, requires_grad=True)
    L = 0.5 * (torch.nn.functional.mse_loss(a - b.mean(), torch.ones(37)) + torch.nn.functional.mse_loss(b - a.mean(), torch.zeros(37)))
    L.backward()
    ab = torch.zeros((2, 37, 32), device="cuda")
    ab[0, :, 0], ab[1, :, 0] = a.detach().cuda(), b.detach().cuda()
    gab, acc = torch.zeros_like(ab), torch.zeros(1, device="cuda")
    _lib.call("fo_ralsgan", ops._ptr(ab[0]), 37, ops._ptr(ab[1]), 37, 32, C.c_float(1.0), C.c_float(0.0), C.c_float(0.5), ops._ptr(acc), None,
              ops._ptr(gab[0]), ops._ptr(gab[1]), ops._stream())
    np.testing.assert_allclose(acc.item(), L.item(), rtol=1e-5)
    assert (gab[0, :, 0].cpu() - a.grad).abs().max().item() <= 1e-6 and (gab[1, :, 0].cpu() - b.grad).abs().max().item() <= 1e-6


@pytest.mark.parametrize("tag,dims", [("v", 3), ("i", 2)])
def test_module_mirror_runs_the_reference_discriminator_update(golden_dir, tag, dims):
    """The reference's own lines (train_vqvae_mocoganhd_disc.py:392-408,428-432) on the mirrored classes through the
    reference's import paths: D(fake), D(real), Relativistic_Average_LSGAN both ways, optim.zero_grad / backward / step --
    losses, parameter gradients, the gradient reaching the fake input, and the updated parameters against the goldens."""
    from TemporalAlignment.models.mocoganhd_video_disc import ModelD_3d
    from TemporalAlignment.models.mocoganhd_content_disc import ModelD_img
    from TemporalAlignment.models import mocoganhd_losses
    g = np.load(os.path.join(golden_dir, "disc_kat.npz"))
    real, fake, F = _inputs(g, tag, dims)
    m = (ModelD_3d(nc=3, norm_D_3d="instance", num_D=2, lr=1e-4, cross_domain=False, n_frames_G=F) if dims == 3
         else ModelD_img(nc=3, norm_D_3d="instance", num_D=2, lr=1e-4)).to("cuda")
    sd = make_disc_state(int(g[f"{tag}_seed_w"]), dims)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert list(m.state_dict()) == list(sd)
    m.train()
    fake = fake.cuda().requires_grad_(True)
    D_fake = m(fake)
    D_real = m(real.cuda())
    assert [tuple(t.shape) for t in D_fake[0]][-1] == g[f"{tag}_fake_s0_logits"].shape
    criterionGAN = mocoganhd_losses.Relativistic_Average_LSGAN()
    D_loss_real = criterionGAN(D_real, D_fake, True)
    D_loss_fake = criterionGAN(D_fake, D_real, False)
    D_loss = (D_loss_real + D_loss_fake) * 0.5
    m.optim.zero_grad()
    D_loss.backward()
    np.testing.assert_allclose([D_loss_real.item(), D_loss_fake.item()], [g[f"{tag}_d_loss_real"], g[f"{tag}_d_loss_fake"]], rtol=1e-3)
    names = [str(n) for n in g[f"{tag}_param_names"]]
    params = dict(m.named_parameters())
    got = np.concatenate([_sub(params[n].grad) for n in names])
    want = g[f"{tag}_d_grad_sub"]
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
    wsub = g[f"{tag}_d_gfake_sub"]
    assert np.abs(_sub(fake.grad) - wsub).max() <= 1e-3 * np.abs(wsub).max()
    m.optim.step()
    after = np.concatenate([_sub(params[n]) for n in names])
    big = np.abs(want) > 1e-3 * np.abs(want).max()
    assert np.abs(after - g[f"{tag}_param_after_sub"])[big].max() <= 5e-6


@pytest.mark.parametrize("Cc,dims", [(128, (5, 33, 31)), (256, (3, 17, 19)), (512, (4, 9, 10))])
def test_chunked_instnorm_vs_torch(Cc, dims):
    """fo_instnorm_lrelu_{fwd,bwd}_batch with a workspace (csrc/disc_ops.hip, round 6: per-chunk (mean, M2) -> fixed-order merge -> elementwise pass) at
    the discriminators' channel counts, ragged row counts (the last chunk is partial), two samples whose running-statistics updates are applied in
    REVERSED order, a large mean against the spread (what E[x^2] - mean^2 would lose): against nn.InstanceNorm3d + LeakyReLU on the CPU, against the
    one-launch kernels (same arithmetic up to summation order), and bit-reproducible."""
    from faceoff_amd import _lib, ops
    g = torch.Generator().manual_seed(Cc)
    N, rows = 2, dims[0] * dims[1] * dims[2]
    x = (torch.randn((N, Cc) + dims, generator=g) * 0.5 + 3.0).requires_grad_(True)
    inorm = torch.nn.InstanceNorm3d(Cc, affine=False, track_running_stats=True)
    outs = [None, None]
    for n in (1, 0):                                                   # module calls: sample 1 first
        outs[n] = torch.nn.functional.leaky_relu(inorm(x[n:n + 1]), 0.2)
    yy = torch.cat(outs)
    gyy = torch.randn(yy.shape, generator=g)
    yy.backward(gyy)
    cl = lambda t: t.detach().permute(0, 2, 3, 4, 1).reshape(N, rows, Cc).contiguous().cuda()
    xc, gc = cl(x), cl(gyy)
    order = torch.tensor([1, 0], dtype=torch.int32, device="cuda")
    nb = int(_lib.load().fo_instnorm_ws_bytes(N, C.c_int64(rows), Cc))
    assert nb > 0
    res = []
    for use_ws in (True, True, False):
        ws = torch.empty(nb // 4, device="cuda") if use_ws else None
        yc, st, gx = torch.empty_like(xc), torch.empty((N, 2 * Cc), device="cuda"), torch.empty_like(xc)
        run = torch.cat([torch.zeros(Cc), torch.ones(Cc)]).cuda()
        _lib.call("fo_instnorm_lrelu_fwd_batch", ops._ptr(xc), Cc, ops._ptr(yc), Cc, N, C.c_int64(rows), Cc, C.c_float(1e-5), C.c_float(0.2), ops._ptr(st),
                  ops._ptr(run), ops._ptr(order), C.c_float(0.1), 0, ops._ptr(ws), C.c_int64(nb if use_ws else 0), ops._stream())
        _lib.call("fo_instnorm_lrelu_bwd_batch", ops._ptr(gc), Cc, ops._ptr(yc), Cc, ops._ptr(st), ops._ptr(gx), Cc, N, C.c_int64(rows), Cc, C.c_float(0.2),
                  ops._ptr(ws), C.c_int64(nb if use_ws else 0), ops._stream())
        torch.cuda.synchronize()
        res.append((yc.cpu(), st.cpu(), run.cpu(), gx.cpu()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)                                       # bit-reproducible
    want, wantg = cl(yy).cpu(), cl(x.grad).cpu()
    for yc, st, run, gx in (res[0], res[2]):
        assert (yc - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        np.testing.assert_allclose(run[:Cc].numpy(), inorm.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(run[Cc:].numpy(), inorm.running_var.numpy(), rtol=2e-5)
        assert (gx - wantg).abs().max().item() <= 5e-5 * wantg.abs().max().item()
    assert (res[0][0] - res[2][0]).abs().max().item() <= 1e-5 * want.abs().max().item()       # chunked vs one-launch kernels
