"""The drop-in nn.Module API (faceoff_amd.models.vqvae_conv3d_latent) on a real MI355X, driven exactly
like the reference trainer drives its model (train_faceoff_perceptual.py:32-47,93-107): torch losses,
loss.backward() through torch autograd, torch.optim.Adam -- against the reference's golden outputs."""
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, golden_state

pytestmark = pytest.mark.gpu
SUB = 61


def _sub(t):
    return t.detach().reshape(-1)[::SUB].cpu().numpy()


def test_reference_style_training_step(golden_dir):
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    g = np.load(os.path.join(golden_dir, "c1_e2e.npz"))
    B, T, H, W = (int(g[k]) for k in "BTHW")
    model = VQVAE(in_channel=3 * 2).to("cuda")                       # utils.py:52
    sd = golden_state(g)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    img, gt = make_batch(int(g["seed_x"]), B, T, H, W)
    img = torch.from_numpy(img).cuda()                               # [B,T,6,H,W]
    gt = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
    optimizer = torch.optim.Adam(model.parameters(), lr=3e-4)        # :190
    criterion = torch.nn.MSELoss()                                   # :21
    model.zero_grad()
    out, latent_loss = model(img)
    assert out.shape == (B * T, 6, H, W) and latent_loss.shape == (1,)
    recon_loss = criterion(out[:, :3], gt)
    loss = recon_loss + 1 * latent_loss.mean()                       # :98 (perceptual term: see test_lpips)
    loss.backward()
    np.testing.assert_allclose(recon_loss.item(), float(g["recon"]), rtol=1e-3)
    np.testing.assert_allclose(latent_loss.item(), float(g["latent"]), rtol=1e-3)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["dec"], rtol=0, atol=1e-3 * np.abs(g["dec"]).max())
    names = [str(n) for n in g["param_names"]]
    params = dict(model.named_parameters())
    assert list(params) == names
    got = np.concatenate([_sub(params[n].grad) for n in names])
    scale = np.abs(g["grad_sub"]).max()
    assert np.abs(got - g["grad_sub"]).max() <= 1e-3 * scale
    for n in names:
        if "grad_full." + n in g.files:
            np.testing.assert_allclose(params[n].grad.cpu().numpy(), g["grad_full." + n], rtol=0,
                                       atol=1e-3 * np.abs(g["grad_full." + n]).max())
    optimizer.step()                                                 # :107
    after = np.concatenate([_sub(params[n]) for n in names])
    np.testing.assert_allclose(after, g["param_after_sub"], rtol=1e-3, atol=3e-5)
    # EMA buffers were updated in place (:66-75) and are what state_dict() saves (:143)
    sd2 = model.state_dict()
    for k in ("quantize_t.embed", "quantize_b.cluster_size", "quantize_b.embed_avg"):
        np.testing.assert_allclose(float(sd2[k].double().pow(2).sum()), g["buf_stats." + k][1], rtol=2e-3)
    # eval-mode forward after the step, no EMA, no grad
    model.eval()
    with torch.no_grad():
        out2, diff2 = model(img)
    np.testing.assert_allclose(diff2.item(), float(g["diff2"].reshape(-1)[0]), rtol=5e-2)
    # inference entry points: codes -> decode_code reproduces the eval forward (:287-295)
    id_t, id_b = model._last_ids
    dec3 = model.decode_code(id_t, id_b)
    assert (dec3 - out2).abs().max().item() <= 1e-4 * out2.abs().max().item()
    enc_b, enc_t = model.only_encode(img.reshape(B * T, 6, H, W))
    assert enc_b.shape == (B * T, 128, H // 4, W // 4) and enc_t.shape == (B * T, 128, H // 8, W // 8)
    assert enc_b.min().item() >= 0.0                                  # Encoder ends in ReLU (:126)
    # checkpoint round trip through the reference's format: bare state_dict, 'module.' prefix stripped on load
    m2 = VQVAE(in_channel=6).to("cuda")
    m2.load_state_dict({"module." + k: v.cpu() for k, v in sd2.items()})
    m2.eval()
    with torch.no_grad():
        out4, _ = m2(img)
    assert torch.equal(out4, out2)


def test_quantize_module_against_reference_kat(golden_dir):
    from faceoff_amd.models.vqvae_conv3d_latent import Quantize
    g = np.load(os.path.join(golden_dir, "quantize_kat.npz"))
    for mode in ("train", "eval"):
        q = Quantize(64, 512).cuda()
        cs0 = torch.from_numpy(g["cluster_size0"]).cuda()
        q.embed.copy_(torch.from_numpy(g["embed"]))
        q.embed_avg.copy_(q.embed * cs0[None, :])
        q.cluster_size.copy_(cs0)
        q.train(mode == "train")
        x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
        quant, diff, ind = q(x)
        (quant * torch.from_numpy(g[f"{mode}_gout"]).cuda()).sum().add(diff * 3.0).backward()
        assert np.array_equal(ind.cpu().numpy().astype(np.int16), g[f"{mode}_ind"])
        np.testing.assert_allclose(quant.detach().cpu().numpy(), g[f"{mode}_quantize"], rtol=1e-6)
        np.testing.assert_allclose(diff.item(), float(g[f"{mode}_diff"]), rtol=1e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{mode}_gx"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(q.embed.cpu().numpy(), g[f"{mode}_embed_after"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(q.cluster_size.cpu().numpy(), g[f"{mode}_cluster_size_after"], rtol=1e-6)
