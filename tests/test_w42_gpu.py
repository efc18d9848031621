"""Winograd F(4x4, 2x2) form of the k4 s2 p1 stems (csrc/wino42.hip; reference models/vqvae_conv3d_latent.py:108,110,117,157,160,215)
against torch-CPU fp32 convolutions of the same operands: forward (with bias / ReLU / channel-slice output), the transposed form
(ConvTranspose2d forward = Conv2d data gradient, with mask / add) and the filter gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5      # of the tensor scale (fp32 Winograd F(4x4, 2x2): observed ~1e-6)


def _close(got, want, tol=TOL):
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item() / scale
    assert err <= tol, err
    return err


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


@pytest.mark.parametrize("cin,cout,hw", [(64, 128, (64, 64)), (32, 128, (48, 80)), (16, 256, (64, 48))])
def test_conv_k4s2_winograd_vs_torch(cin, cout, hw, monkeypatch):
    from faceoff_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    N, (H, W) = 24, hw
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 4, 4), generator=g) / np.sqrt(16 * cin)
    b = torch.randn(cout, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=2, padding=1))
    assert ops.w42_conv_ok(N, H, W, cin, cout)
    U = ops.w42_filter(w.cuda(), False)
    out = torch.full((N, H // 2, W // 2, cout + 32), 7.0, device="cuda")
    V = ops.conv_k4s2_winograd(_nhwc(x), U, b.cuda(), out[..., :cout], cin=cin, cout=cout, flags=ops.FO_OUT_RELU, keep_v=True)
    err = _close(out[..., :cout].cpu().permute(0, 3, 1, 2), ref)
    assert (out[..., cout:] == 7.0).all()
    # filter gradient, with the kept V and without
    gy = torch.randn((N, cout, H // 2, W // 2), generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, None, stride=2, padding=1).backward(gy)
    if ops.w42_wgrad_ok(N, H, W, cin, cout):
        for keep in (V, None):
            dw = torch.empty_like(w, device="cuda")
            ops.conv_k4s2_wgrad_winograd(_nhwc(x), _nhwc(gy), dw, cin=cin, cout=cout, V=keep)
            _close(dw.cpu(), wr.grad, 5e-5)
    # data gradient = the transposed form
    if ops.w42_convT_ok(N, H // 2, W // 2, cout, cin):
        Ut = ops.w42_filter(w.cuda(), True)
        gx = torch.empty((N, H, W, cin), device="cuda")
        ops.convT_k4s2_winograd(_nhwc(gy), Ut, None, gx, cin=cout, cout=cin)
        _close(gx.cpu().permute(0, 3, 1, 2), xr.grad)
    print(f"[w42 conv {cin}->{cout} {hw}] forward rel err {err:.2e}")


@pytest.mark.parametrize("cin,cout,hw", [(128, 64, (32, 32)), (64, 64, (24, 40)), (128, 32, (21, 37))])
def test_convT_k4s2_winograd_vs_torch(cin, cout, hw):
    from faceoff_amd import ops
    g = torch.Generator().manual_seed(cin * 3 + cout)
    N, (h, w_) = 20, hw
    x = torch.randn((N, cin, h, w_), generator=g)
    w = torch.randn((cin, cout, 4, 4), generator=g) / np.sqrt(4 * cin)
    b = torch.randn(cout, generator=g)
    mask = torch.randn((N, cout, 2 * h, 2 * w_), generator=g)
    add = torch.randn((N, cout, 2 * h, 2 * w_), generator=g)
    y = torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=1)
    ref = y * (mask > 0) + add
    assert ops.w42_convT_ok(N, h, w_, cin, cout)
    U = ops.w42_filter(w.cuda(), True)
    out = torch.empty((N, 2 * h, 2 * w_, cout), device="cuda")
    ops.convT_k4s2_winograd(_nhwc(x), U, b.cuda(), out, cin=cin, cout=cout, mask=_nhwc(mask), add=_nhwc(add))
    err = _close(out.cpu().permute(0, 3, 1, 2), ref)
    out2 = torch.empty((N, 2 * h, 2 * w_, cout), device="cuda")
    ops.convT_k4s2_winograd(_nhwc(x), U, b.cuda(), out2, cin=cin, cout=cout, flags=ops.FO_OUT_RELU)
    _close(out2.cpu().permute(0, 3, 1, 2), torch.relu(y))
    print(f"[w42 convT {cin}->{cout} {hw}] rel err {err:.2e}")
