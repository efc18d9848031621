"""LPIPS / VGG-16 perceptual loss on a real MI355X against the reference's golden output
(tests/golden/lpips_kat.npz: reference loss.VQLPIPS with torchvision shimmed and seeded weights) and the
CPU oracle at a ragged size.  Pretrained-weight parity is unpinned (weights are a network download)."""
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_vgg_lpips_state

pytestmark = pytest.mark.gpu


def test_vqlpips_module_vs_reference_golden(golden_dir):
    from faceoff_amd.loss import VQLPIPS
    g = np.load(os.path.join(golden_dir, "lpips_kat.npz"))
    m = VQLPIPS(make_vgg_lpips_state(int(g["seed"]))).cuda()
    assert not any(p.requires_grad for p in m.parameters())
    tgt = torch.from_numpy(g["target"]).cuda()
    rec = torch.from_numpy(g["recon"]).cuda().requires_grad_(True)
    val = m(tgt, rec)                                   # loss.py:33
    assert val.dim() == 0
    np.testing.assert_allclose(val.item(), float(g["value"]), rtol=1e-3)
    np.testing.assert_allclose(m._engine.last_per_image.cpu().numpy(), g["per_image"].reshape(-1), rtol=1e-3)
    (val * 2.0).backward()
    want = 2.0 * g["grad_recon"]
    got = rec.grad.cpu().numpy()
    # Max-pool arg-max is discontinuous: this fixture has a 2x2 window whose two largest relu3_3 values differ by
    # 3 ulp (5.2393827 vs 5.2393842), so a different fp32 summation order inside the conv legitimately routes that
    # window's gradient to the other pixel (like the VQ near-ties).  Hence: 1e-3 for all but a receptive field's
    # worth of pixels, and a tight bound on the relative L2 error of the whole gradient.
    bad = np.abs(got - want) > 1e-3 * np.abs(want).max()
    assert bad.mean() < 0.05, bad.mean()
    assert np.linalg.norm(got - want) <= 2e-2 * np.linalg.norm(want)


def test_lpips_trainer_path_vs_oracle_ragged():
    """Fast path (gradient accumulated into the NHWC decoder-output gradient) at 3 x 48x80 against the oracle."""
    from faceoff_amd.lpips import LPIPSEngine
    from oracle import faceoff_oracle as O
    sd = make_vgg_lpips_state(3)
    rng = np.random.default_rng(8)
    tgt = rng.uniform(-1, 1, (3, 3, 48, 80)).astype(np.float32)
    rec = (tgt + 0.4 * rng.standard_normal(tgt.shape)).astype(np.float32)
    lp = {k: torch.from_numpy(v) for k, v in sd.items()}
    r = torch.from_numpy(rec).requires_grad_(True)
    ref = O.lpips_forward(torch.from_numpy(tgt), r, lp).mean()
    (0.5 * ref).backward()
    eng = LPIPSEngine(sd, "cuda:0")
    dec = torch.zeros((3, 48, 80, 8), device="cuda")
    dec[..., :3] = torch.from_numpy(rec).permute(0, 2, 3, 1).cuda()
    base = torch.from_numpy((1e-6 * rng.standard_normal((3, 48, 80, 8))).astype(np.float32)).cuda()
    g_dec = base.clone()                                # pre-existing gradient (the MSE term) must be kept
    loss = eng.loss_and_grad(torch.from_numpy(tgt).cuda(), dec, g_dec, weight=0.5)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-3)
    got = (g_dec[..., :3] - base[..., :3]).permute(0, 3, 1, 2).cpu().numpy()
    want = r.grad.numpy()
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    assert torch.equal(g_dec[..., 3:], base[..., 3:])


def test_lpips_frame_chunking_is_transparent():
    """Batches whose relu1_2 map exceeds the conv kernel's 2 GiB window run in frame chunks: same loss, same gradient."""
    from faceoff_amd.lpips import LPIPSEngine
    eng = LPIPSEngine(make_vgg_lpips_state(5), "cuda:0")
    rng = np.random.default_rng(2)
    tgt = torch.from_numpy(rng.uniform(-1, 1, (5, 3, 32, 32)).astype(np.float32)).cuda()
    dec = torch.zeros((5, 32, 32, 8), device="cuda")
    dec[..., :3] = tgt.permute(0, 2, 3, 1) + 0.3 * torch.from_numpy(rng.standard_normal((5, 32, 32, 3)).astype(np.float32)).cuda()
    g1, g2 = torch.zeros_like(dec), torch.zeros_like(dec)
    l1 = eng.loss_and_grad(tgt, dec, g1)
    eng.window_bytes = 2 * 32 * 32 * 64 * 4              # two frames per chunk -> chunks of 2, 2, 1
    l2 = eng.loss_and_grad(tgt, dec, g2)
    np.testing.assert_allclose(l2.item(), l1.item(), rtol=1e-6)
    assert (g1 - g2).abs().max().item() <= 1e-6 * g1.abs().max().item()
    assert eng.last_per_image.shape == (5,)
