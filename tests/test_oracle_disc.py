"""The discriminator oracle (oracle/disc_oracle.py) against golden vectors produced by the reference's own ModelD_3d /
ModelD_img / Relativistic_Average_LSGAN (tests/golden/make_golden_disc.py): multiscale logits, both loss forms, gradients
with respect to every parameter and to the fake input, InstanceNorm running statistics, one Adam(0.5, 0.999) step."""
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_disc_state
from oracle import disc_oracle as D

SUB = 211


def _sub(t):
    return t.detach().reshape(-1)[::SUB].numpy()


def _stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.pow(2).sum().item(), t.abs().max().item()])


def _inputs(g, tag, dims):
    F, H, W = (int(g[f"{tag}_{k}"]) for k in "FHW")
    rng = np.random.default_rng(int(g[f"{tag}_seed_x"]))
    shape = (1, 6, F - 1, H, W) if dims == 3 else (1, 6, H, W)
    real = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    return real, fake, F


@pytest.mark.parametrize("tag,dims", [("v", 3), ("i", 2)])
def test_discriminator_oracle_vs_reference_golden(golden_dir, tag, dims):
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "disc_kat.npz"))
    real, fake, F = _inputs(g, tag, dims)
    sd = make_disc_state(int(g[f"{tag}_seed_w"]), dims)
    p = D.to_torch_state(sd)
    fake.requires_grad_(True)
    bufs = {}
    Df = D.multiscale_discriminator(fake, p, n_frames=F - 1, buffers=bufs)
    Dr = D.multiscale_discriminator(real, p, n_frames=F - 1, buffers=bufs)
    for s in range(2):
        for got_l, key in ((Df[s][-1], f"{tag}_fake_s{s}_logits"), (Dr[s][-1], f"{tag}_real_s{s}_logits")):
            assert np.abs(got_l.detach().numpy() - g[key]).max() <= 2e-5 * np.abs(g[key]).max(), key
        for j in range(5):
            np.testing.assert_allclose(_stats(Df[s][j]), g[f"{tag}_fake_s{s}_l{j}_stats"], rtol=1e-4)
    d_real, d_fake = D.ralsgan(Dr, Df, True), D.ralsgan(Df, Dr, False)
    d_loss = (d_real + d_fake) * 0.5
    np.testing.assert_allclose([d_real.item(), d_fake.item(), d_loss.item()],
                               [g[f"{tag}_d_loss_real"], g[f"{tag}_d_loss_fake"], g[f"{tag}_d_loss"]], rtol=1e-5)
    d_loss.backward()
    names = [str(n) for n in g[f"{tag}_param_names"]]
    assert names == [k for k, v in p.items() if v.requires_grad]
    gs = np.stack([_stats(p[n].grad) for n in names])
    # (the bias gradients in front of an InstanceNorm are zero up to rounding: compare on the scale of the largest tensor)
    np.testing.assert_allclose(gs[:, 1], g[f"{tag}_d_grad_stats"][:, 1], rtol=1e-3, atol=1e-9 * g[f"{tag}_d_grad_stats"][:, 1].max())
    got = np.concatenate([_sub(p[n].grad) for n in names])
    want = g[f"{tag}_d_grad_sub"]
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    np.testing.assert_allclose(_sub(fake.grad), g[f"{tag}_d_gfake_sub"], rtol=1e-3, atol=1e-5 * float(np.abs(g[f"{tag}_d_gfake_sub"]).max()))
    for k, b in bufs.items():                     # running statistics after the two module calls (fake, then real)
        np.testing.assert_allclose(b.numpy(), g[f"{tag}_buf.{k}"], rtol=1e-4, atol=1e-6)
    # one Adam(betas=(0.5, 0.999), lr=1e-4) step: where the gradient is not rounding noise, the update is lr * sign(g)
    grads = {n: p[n].grad.clone() for n in names}
    before = np.concatenate([_sub(p[n]) for n in names])
    D.adam_step(p, grads, {}, lr=1e-4, betas=(0.5, 0.999))
    after = np.concatenate([_sub(p[n]) for n in names])
    big = np.abs(want) > 1e-3 * np.abs(want).max()
    assert np.abs(after - g[f"{tag}_param_after_sub"])[big].max() <= 2e-6
    assert np.abs(after - before).max() <= 1.01e-4
    # generator-style loss: gradient with respect to the fake input
    p2 = D.to_torch_state(sd)
    fake2 = fake.detach().clone().requires_grad_(True)
    Df2, Dr2 = D.multiscale_discriminator(fake2, p2, n_frames=F - 1), D.multiscale_discriminator(real, p2, n_frames=F - 1)
    g_loss = (D.ralsgan(Df2, Dr2, True) + D.ralsgan(Dr2, Df2, False)) * 0.5
    g_loss.backward()
    np.testing.assert_allclose(g_loss.item(), float(g[f"{tag}_g_loss"]), rtol=1e-5)
    np.testing.assert_allclose(_sub(fake2.grad), g[f"{tag}_g_gfake_sub"], rtol=1e-3, atol=1e-5 * float(np.abs(g[f"{tag}_g_gfake_sub"]).max()))


def test_frame_pairing_matches_the_trainer_expression():
    """pair_with_first == cat((x[:,0].unsqueeze(1).repeat(1,F-1,1,1,1), x[:,1:]), dim=2).transpose(1,2) (trainer :395-401)."""
    x = torch.arange(2 * 5 * 3 * 2 * 2, dtype=torch.float32).reshape(2, 5, 3, 2, 2)
    want = torch.cat((x[:, 0].unsqueeze(1).repeat(1, 4, 1, 1, 1), x[:, 1:]), dim=2).transpose(1, 2)
    assert torch.equal(D.pair_with_first(x), want)
    assert torch.equal(D.flip_video(want, True), torch.flip(want, [2])) and D.flip_video(want, False) is want
    assert torch.equal(D.image_pair(x, 3), torch.cat((x[:, 0], x[:, 3]), dim=1))
