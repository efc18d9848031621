"""BASELINE config 5 on the MI355X: the generator and the discriminator iteration of the two-optimiser GAN loop
(disc_trainers/train_vqvae_mocoganhd_disc.py:303-432) -- VQ-VAE generator, MoCoGAN-HD video + image discriminators,
relativistic average LSGAN -- against the CPU oracle's restatement of the same iterations (every block of which is pinned by
reference goldens: tests/test_oracle_golden.py, tests/test_oracle_disc.py).  Random choices are fixed arguments."""
import random

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, make_disc_state

pytestmark = pytest.mark.gpu
N, H, W, WIN = 8, 32, 32, 6


def _setup():
    from faceoff_amd.disc import DiscEngine
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.gan_trainer import GANTrainer
    sd, sd3, sd2 = make_state_dict(2, codebook_scale=0.3, gain=2.0), make_disc_state(8, 3), make_disc_state(9, 2)
    img, gt = make_batch(77, 1, N, H, W)
    eng = VQVAEEngine(sd, "cuda:0")
    d3 = DiscEngine(sd3, "cuda:0", dims=3, n_frames=WIN - 1)
    d2 = DiscEngine(sd2, "cuda:0", dims=2)
    tr = GANTrainer(eng, d3, d2, lr=3e-4, d_lr=1e-4, window=WIN)
    return sd, sd3, sd2, img, gt, eng, d3, d2, tr


def _oracle_forward(sd, img, gt, r):
    from oracle import faceoff_oracle as O
    p = O.to_torch_state(sd)
    fw = O.vqvae_forward(torch.from_numpy(img), p, training=True)
    out = fw["dec"][:, :3]
    gtt = torch.from_numpy(gt).reshape(N, 3, H, W)
    recon = torch.nn.functional.mse_loss(out, gtt)
    latent = fw["diff"].mean()
    return p, fw, recon, latent, out[r:r + WIN].unsqueeze(0), gtt[r:r + WIN].unsqueeze(0)


def _worst(got, want):
    worst = (0.0, "")
    for k, w in want.items():
        scale = max(w.abs().max().item(), w.pow(2).mean().sqrt().item()) + 1e-30
        worst = max(worst, (float((got[k].cpu() - w).abs().max().item() / scale), k))
    return worst


def test_generator_iteration_vs_oracle():
    from oracle import disc_oracle as D
    sd, sd3, sd2, img, gt, eng, d3, d2, tr = _setup()
    c = dict(random_idx=1, frame_id=3, flip_real=True, flip_fake=False)
    p, fw, recon, latent, x_fake, x = _oracle_forward(sd, img, gt, c["random_idx"])
    p3, p2, b3, b2 = D.to_torch_state(sd3), D.to_torch_state(sd2), {}, {}
    g2d, g3d = D.generator_gan_losses(x_fake, x, p3, p2, c["frame_id"], c["flip_real"], c["flip_fake"], b3, b2)
    (recon + latent + g2d + g3d).backward()                                   # G_loss (:375)
    tr.optimizer.step = lambda grad_scale=1.0: None                           # keep the gradients for the comparison
    o = tr.step(torch.from_numpy(img).reshape(N, 6, H, W).cuda(), torch.from_numpy(gt).reshape(N, 3, H, W).cuda(), c)
    torch.cuda.synchronize()
    np.testing.assert_allclose([o["recon"].item(), o["latent"].item(), o["g_loss_2d"].item(), o["g_loss_3d"].item()],
                               [recon.item(), latent.item(), g2d.item(), g3d.item()], rtol=1e-3)
    worst = _worst(eng.grads, {k: v.grad for k, v in p.items() if v.requires_grad})
    assert worst[0] <= 1e-3, worst
    for eng_d, bufs in ((d3, b3), (d2, b2)):                                  # running statistics moved in the reference's call order
        sdd = eng_d.state_dict()
        for k, v in bufs.items():
            assert (sdd[k].cpu() - v).abs().max().item() <= 1e-3 * max(v.abs().max().item(), 1e-3), k
    assert tr.iteration == 1
    print(f"[GAN generator iteration] G_2d {o['g_loss_2d'].item():.6f} G_3d {o['g_loss_3d'].item():.6f}; worst generator-gradient rel err {worst}")


def test_discriminator_iteration_vs_oracle():
    from oracle import disc_oracle as D
    sd, sd3, sd2, img, gt, eng, d3, d2, tr = _setup()
    tr.iteration = 1                                                          # odd: the discriminator branch (:338-341)
    c = dict(random_idx=2, frame_id=4, flip_real=False, flip_fake=True)
    p, fw, recon, latent, x_fake, x = _oracle_forward(sd, img, gt, c["random_idx"])
    p3, p2 = D.to_torch_state(sd3), D.to_torch_state(sd2)
    dl3, dl2 = D.discriminator_losses(x_fake, x, p3, p2, c["frame_id"], c["flip_real"], c["flip_fake"], {}, {})
    dl3.backward()
    dl2.backward()
    params_before = eng.flat_params.clone()
    o = tr.step(torch.from_numpy(img).reshape(N, 6, H, W).cuda(), torch.from_numpy(gt).reshape(N, 3, H, W).cuda(), c)
    torch.cuda.synchronize()
    np.testing.assert_allclose([o["d_loss_3d"].item(), o["d_loss_2d"].item()], [dl3.item(), dl2.item()], rtol=1e-3)
    assert torch.equal(eng.flat_params, params_before)                        # the generator does not move on this iteration
    for eng_d, pd in ((d3, p3), (d2, p2)):
        want = {k: v.grad for k, v in pd.items() if v.requires_grad}
        tot = max(w.abs().max().item() for w in want.values())
        for k, w in want.items():
            g = eng_d.grads[k].cpu()
            if w.abs().max().item() < 1e-4 * tot:                             # bias in front of an InstanceNorm: zero up to rounding
                assert g.abs().max().item() <= 2e-4 * tot, k
                continue
            assert (g - w).abs().max().item() <= 1e-3 * w.abs().max().item(), k
        grads = {k: w.clone() for k, w in want.items()}
        D.adam_step(pd, grads, {}, lr=1e-4, betas=(0.5, 0.999))
        for k, w in want.items():                                             # the update, where the gradient is not rounding noise
            if w.abs().max().item() < 1e-4 * tot:
                continue
            big = w.abs() > 1e-3 * w.abs().max()
            if big.any():
                assert (eng_d.params[k].cpu() - pd[k].detach())[big].abs().max().item() <= 5e-6, k
    assert tr.iteration == 2


def test_random_choices_follow_the_reference_call_order():
    """GANTrainer.draw consumes `random` exactly as the reference's branches do (:330,351,368-369 / :330,392-393,411)."""
    from faceoff_amd.gan_trainer import GANTrainer
    tr = GANTrainer.__new__(GANTrainer)
    tr.window, tr.rng = 16, random.Random(5)
    ref = random.Random(5)
    g = tr.draw(30, True)
    assert g == dict(random_idx=ref.randint(0, 14), frame_id=ref.randint(1, 15), flip_real=ref.randint(0, 1) == 0, flip_fake=ref.randint(0, 1) == 0)
    d = tr.draw(30, False)
    want = dict(random_idx=ref.randint(0, 14))
    want["flip_fake"] = ref.randint(0, 1) == 0
    want["flip_real"] = ref.randint(0, 1) == 0
    want["frame_id"] = ref.randint(1, 15)
    assert d == want


def test_gan_iterations_are_bit_reproducible():
    """Two trainers from the same state, two iterations each (generator, then discriminator), at a size where the discriminators'
    convolutions and filter gradients run sliced (K-slices / row slices through workspaces, added in slice order): every parameter
    of the generator and of both discriminators, the running statistics and the LOSS SCALARS of both iterations equal bit for bit."""
    from faceoff_amd.disc import DiscEngine
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.gan_trainer import GANTrainer
    n, h, w, win = 10, 128, 128, 10
    img, gt = make_batch(5, 1, n, h, w)
    x = torch.from_numpy(img).reshape(n, 6, h, w).cuda()
    y = torch.from_numpy(gt).reshape(n, 3, h, w).cuda()
    states = []
    for _ in range(2):
        eng = VQVAEEngine(make_state_dict(2, codebook_scale=0.3, gain=2.0), "cuda:0")
        d3 = DiscEngine(make_disc_state(8, 3), "cuda:0", dims=3, n_frames=win - 1)
        d2 = DiscEngine(make_disc_state(9, 2), "cuda:0", dims=2)
        tr = GANTrainer(eng, d3, d2, lr=3e-4, d_lr=1e-4, window=win, rng=random.Random(4))
        losses = {}
        for it in range(2):
            losses.update({f"loss{it}." + k: v.clone() for k, v in tr.step(x, y).items()})
        torch.cuda.synchronize()
        st = {"g." + k: v.clone() for k, v in eng.state_dict().items()}
        st.update(losses)              # the printed losses too: MSE / commitment sums as ordered partials, RaLSGAN as ordered launches
        st.update({"d3." + k: v.clone() for k, v in d3.state_dict().items()})
        st.update({"d2." + k: v.clone() for k, v in d2.state_dict().items()})
        states.append(st)
    diff = [k for k in states[0] if not torch.equal(states[0][k], states[1][k])]
    assert not diff, diff[:8]
