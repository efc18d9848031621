"""Drop-in use on a real MI355X through the reference's own import paths: the training iteration of
train_faceoff_perceptual.py (:32-47 run_step, :93-121 loop body) restated here with the reference's unmodified import
lines (top-level `models`, `loss`, `distributed`, `scheduler`, `utils`, `config`), torch autograd and torch.optim.Adam --
against the CPU oracle.  And the same module wrapped in nn.parallel.DistributedDataParallel (:164-169)."""
import os
import tempfile

import numpy as np
import pytest
import torch
from torch import nn, optim

from faceoff_amd.synth import make_state_dict, make_batch, make_vgg_lpips_state

pytestmark = pytest.mark.gpu


def _loader_tuple(seed, T, H, W):
    """What the reference's DataLoader(batch_size=1) yields: five [1,T,3,H,W] tensors (dataset.py:356-375)."""
    rng = np.random.default_rng(seed)
    return tuple(torch.from_numpy(rng.uniform(-1, 1, (1, T, 3, H, W)).astype(np.float32)) for _ in range(5))


def test_training_iteration_through_the_reference_import_paths():
    from scheduler import CycleScheduler                      # train_faceoff_perceptual.py:14
    import distributed as dist                                # :15
    from utils import process_data                            # :17 (star import in the reference)
    from config import LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT   # :18
    from models.vqvae_conv3d_latent import VQVAE              # utils.py:47
    from loss import VQLPIPS                                  # utils.py:48
    from oracle import faceoff_oracle as O
    device = "cuda"
    T, H, W = 3, 64, 64
    sd = make_state_dict(2, codebook_scale=0.3, gain=2.0)
    lp = make_vgg_lpips_state(7)
    model = VQVAE(in_channel=3 * 2).to(device)                # utils.py:52
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    vqlpips = VQLPIPS().to(device)                            # utils.py:53 (weights: loaded, not downloaded)
    vqlpips.load_state_dict(lp)
    optimizer = optim.Adam(model.parameters(), lr=3e-4)       # :190
    scheduler = CycleScheduler(optimizer, 3e-4, n_iter=40, momentum=None, warmup_proportion=0.05)   # :194-201
    criterion = nn.MSELoss()                                  # :21
    data = _loader_tuple(9, T, H, W)
    model.train()
    # ---- the loop body (:93-121)
    model.zero_grad()
    img, S, ground_truth, _ = process_data(data, device, None)
    out, latent_loss = model(img)
    out = out[:, :3]
    recon_loss = criterion(out, ground_truth)
    latent_loss = latent_loss.mean()
    perceptual_loss = vqlpips(ground_truth, out)
    loss = recon_loss + LATENT_LOSS_WEIGHT * latent_loss + PERCEPTUAL_LOSS_WEIGHT * perceptual_loss
    loss.backward()
    scheduler.step()
    optimizer.step()
    gathered = dist.all_gather({"mse_sum": recon_loss.item() * S, "mse_n": S})
    assert S == T and gathered == [{"mse_sum": recon_loss.item() * S, "mse_n": S}] and dist.is_primary()
    # ---- checker: the CPU oracle on the same clip, same schedule
    p = O.to_torch_state(sd)
    x = torch.cat([data[0], data[2]], dim=2)                  # [1,T,6,H,W]
    lr1 = CycleScheduler(type("o", (), {"param_groups": [{"lr": 0.0}]})(), 3e-4, n_iter=40, momentum=None, warmup_proportion=0.05).step()[0]
    r = O.train_step(x, data[3], p, lpips_state={k: torch.from_numpy(v) for k, v in lp.items()}, adam_state={}, lr=lr1)
    np.testing.assert_allclose(recon_loss.item(), r["recon"].item(), rtol=1e-3)
    np.testing.assert_allclose(latent_loss.item(), r["latent"].item(), rtol=1e-3)
    np.testing.assert_allclose(perceptual_loss.item(), r["perceptual"].item(), rtol=1e-3)
    worst = 0.0
    for k, v in model.named_parameters():
        g = r["grads"][k]
        err = (v.grad.cpu() - g).abs().max().item() / (g.abs().max().item() + 1e-30)
        worst = max(worst, err)
        assert err <= 1e-3, (k, err)
        # Adam moves a parameter by ~lr * sign(g): compare the update where the gradient is not rounding noise
        big = g.abs() > 1e-3 * g.abs().max()
        assert (v.detach().cpu() - p[k].detach())[big].abs().max().item() <= 0.05 * lr1, k
    for k, b in model.named_buffers():
        assert (b.cpu() - p[k]).abs().max().item() <= 1e-3 * p[k].abs().max().item(), k
    print(f"[drop-in iteration] worst gradient rel err vs oracle {worst:.2e}")


B, T_, H_, W_ = 2, 2, 64, 64


def _ddp_worker(outdir):
    import distributed as dist
    from models.vqvae_conv3d_latent import VQVAE
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    model = VQVAE(in_channel=6).to("cuda")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0)   # train_faceoff_perceptual.py:164-169
    img, gt = make_batch(100, 2, T_, H_, W_)
    out, latent = model(torch.from_numpy(img[rank]).cuda())           # one clip per rank, like the reference's loader
    loss = nn.functional.mse_loss(out[:, :3], torch.from_numpy(gt[rank]).cuda()) + latent.mean()
    loss.backward()
    torch.cuda.synchronize()
    torch.save({"grads": {k: v.grad.cpu() for k, v in model.module.named_parameters()},
                "embed_b": model.module.state_dict()["quantize_b.embed"].cpu()}, os.path.join(outdir, f"rank{rank}.pt"))


def test_module_wrapped_in_torch_ddp_averages_gradients_and_sums_vq_statistics():
    """nn.parallel.DistributedDataParallel(model) (the reference's wrap) works on the drop-in module: DDP's hooks see the
    70 parameter gradients the engine hands to autograd and average them; the in-forward VQ statistics all-reduce
    (vqvae_conv3d_latent.py:63-64) runs through `distributed.all_reduce`.  Two ranks (gloo over CUDA tensors, both on
    cuda:0 -- the box has one GPU) against one process that runs the two clips one after the other."""
    import distributed as dist
    from models.vqvae_conv3d_latent import VQVAE
    with tempfile.TemporaryDirectory() as td:
        dist.launch(_ddp_worker, 2, 1, 0, "auto", args=(td,), backend="gloo")
        r = [torch.load(os.path.join(td, f"rank{i}.pt")) for i in range(2)]
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(100, 2, T_, H_, W_)
    serial = []
    for c in range(2):
        model = VQVAE(in_channel=6).to("cuda")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out, latent = model(torch.from_numpy(img[c]).cuda())
        (nn.functional.mse_loss(out[:, :3], torch.from_numpy(gt[c]).cuda()) + latent.mean()).backward()
        serial.append({k: v.grad.cpu() for k, v in model.named_parameters()})
    for k in serial[0]:
        want = (serial[0][k] + serial[1][k]) / 2
        assert torch.equal(r[0]["grads"][k], r[1]["grads"][k]), k
        assert (r[0]["grads"][k] - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-12, k
    assert torch.equal(r[0]["embed_b"], r[1]["embed_b"])
    # ... and against the CPU ORACLE on the two clips as one batch: DDP's average of two per-clip mean losses is the
    # gradient of the global mean, and the in-forward all-reduce makes the EMA statistics those of both clips (vqvae_conv3d_latent.py:59-64)
    from oracle import faceoff_oracle as O
    p = O.to_torch_state(sd)
    o = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p)
    worst = max(((r[0]["grads"][k] - g).abs().max().item() / (g.abs().max().item() + 1e-30), k) for k, g in o["grads"].items())
    print(f"[DDP-wrapped module, two ranks vs CPU oracle on both clips] worst gradient rel err {worst}")
    assert worst[0] <= 1e-3, worst
    e = p["quantize_b.embed"]
    assert (r[0]["embed_b"] - e).abs().max().item() <= 1e-3 * e.abs().max().item()
