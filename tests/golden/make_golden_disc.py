#!/usr/bin/env python3
"""Golden vectors for the MoCoGAN-HD discriminators, produced by the REFERENCE's own classes (imported from
/root/reference): ModelD_3d, ModelD_img (TemporalAlignment/models/mocoganhd_{video,content}_disc.py) and
Relativistic_Average_LSGAN (mocoganhd_losses.py).  Run in the build container only:

    python tests/golden/make_golden_disc.py        -> tests/golden/disc_kat.npz  (outputs only; inputs / weights are re-created
                                                       anywhere from faceoff_amd.synth.make_disc_state and numpy seeds)
Constructor arguments (not pinned by any caller in the reference tree): nc=3, norm_D_3d='instance', num_D=2,
cross_domain=False, n_frames_G=8 here (a 8-frame window -> 7 frame pairs; the trainer uses 16)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FACEOFF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
from faceoff_amd.synth import make_disc_state  # noqa: E402

torch.set_num_threads(8)
SUB = 211


def sub(t):
    return t.detach().reshape(-1)[::SUB].numpy().copy()


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.pow(2).sum().item(), t.abs().max().item()])


def run(dims, seed_w, seed_x, F, H, W):
    from TemporalAlignment.models.mocoganhd_video_disc import ModelD_3d
    from TemporalAlignment.models.mocoganhd_content_disc import ModelD_img
    from TemporalAlignment.models.mocoganhd_losses import Relativistic_Average_LSGAN
    m = (ModelD_3d(nc=3, norm_D_3d="instance", num_D=2, lr=1e-4, cross_domain=False, n_frames_G=F) if dims == 3
         else ModelD_img(nc=3, norm_D_3d="instance", num_D=2, lr=1e-4))
    sd = make_disc_state(seed_w, dims)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.train()
    rng = np.random.default_rng(seed_x)
    shape = (1, 6, F - 1, H, W) if dims == 3 else (1, 6, H, W)
    real = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32)).requires_grad_(True)
    crit = Relativistic_Average_LSGAN()
    res = {"F": F, "H": H, "W": W, "seed_w": seed_w, "seed_x": seed_x}
    # discriminator-style loss (both logits carry gradient to the parameters), module calls: fake first, then real
    D_fake = m(fake)
    D_real = m(real)
    for s in range(2):
        for j in range(5):
            res[f"fake_s{s}_l{j}_stats"] = stats(D_fake[s][j])
        res[f"fake_s{s}_logits"] = D_fake[s][-1].detach().numpy()
        res[f"real_s{s}_logits"] = D_real[s][-1].detach().numpy()
    d_real, d_fake = crit(D_real, D_fake, True), crit(D_fake, D_real, False)
    d_loss = (d_real + d_fake) * 0.5
    m.zero_grad()
    d_loss.backward()
    res.update(d_loss_real=d_real.item(), d_loss_fake=d_fake.item(), d_loss=d_loss.item())
    names = [k for k, _ in m.named_parameters()]
    res["param_names"] = np.array(names)
    res["d_grad_stats"] = np.stack([stats(p.grad) for _, p in m.named_parameters()])
    res["d_grad_sub"] = np.concatenate([sub(p.grad) for _, p in m.named_parameters()])
    res["d_gfake_stats"] = stats(fake.grad)
    res["d_gfake_sub"] = sub(fake.grad)
    for k, b in m.named_buffers():
        res["buf." + k] = b.numpy().copy()
    # generator-style loss on the same logits (gradient wrt the fake input is what the generator receives)
    fake2 = fake.detach().clone().requires_grad_(True)
    m2 = type(m)(*((3, "instance", 2, 1e-4, False, F) if dims == 3 else (3, "instance", 2, 1e-4)))
    m2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m2.train()
    Df, Dr = m2(fake2), m2(real)
    g_loss = (crit(Df, Dr, True) + crit(Dr, Df, False)) * 0.5
    g_loss.backward()
    res.update(g_loss=g_loss.item(), g_gfake_stats=stats(fake2.grad), g_gfake_sub=sub(fake2.grad),
               g_gfake_full=fake2.grad.numpy() if fake2.grad.numel() < 60000 else np.zeros(0, np.float32))
    # one Adam step of the module's own optimiser (betas 0.5, 0.999) on the discriminator gradients
    m.optim.step()
    res["param_after_sub"] = np.concatenate([sub(p) for _, p in m.named_parameters()])
    return res


if __name__ == "__main__":
    out = {}
    for tag, dims, F, H, W in (("v", 3, 8, 32, 32), ("i", 2, 8, 48, 40)):
        r = run(dims, seed_w=5 + dims, seed_x=50 + dims, F=F, H=H, W=W)
        print(tag, "d_loss", r["d_loss"], "g_loss", r["g_loss"], "logit shapes", r["fake_s0_logits"].shape, r["fake_s1_logits"].shape)
        out.update({f"{tag}_{k}": v for k, v in r.items()})
    np.savez(os.path.join(HERE, "disc_kat.npz"), **out)
