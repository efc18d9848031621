"""Data-parallel semantics on the real GPU path: two ranks (gloo over CUDA tensors, both on cuda:0 because
the GPU box has one device) run FaceOffTrainer with the bucketed side-stream all-reduce and the fused VQ
statistics all-reduce; the result must equal ONE process stepping on the concatenated batch (DDP mean of
per-rank means over equal shards = global mean; VQ statistics are summed; reference
train_faceoff_perceptual.py:164-169, vqvae_conv3d_latent.py:63-64)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from faceoff_amd import distributed as dist
from faceoff_amd.synth import make_state_dict, make_batch

pytestmark = pytest.mark.gpu
B, T, H, W = 2, 2, 64, 64


def _worker(outdir):
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4, bucket_bytes=2 << 20)
    assert tr.reducer is not None and len(tr.reducer.buckets) >= 4
    img, gt = make_batch(100, 2 * B, T, H, W)
    sl = slice(rank * B, (rank + 1) * B)
    recon, latent, _ = tr.step(torch.from_numpy(img[sl]).cuda(), torch.from_numpy(gt[sl]).cuda())
    torch.cuda.synchronize()
    torch.save({"params": eng.flat_params.cpu(), "grads": eng.flat_grads.cpu(),
                "embed_b": eng.buffers["quantize_b.embed"].cpu(), "recon": recon.cpu(), "latent": latent.cpu()},
               os.path.join(outdir, f"rank{rank}.pt"))


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    with tempfile.TemporaryDirectory() as td:
        dist.launch(_worker, 2, 1, 0, "auto", args=(td,), backend="gloo")
        r = [torch.load(os.path.join(td, f"rank{i}.pt")) for i in range(2)]
    # both ranks hold identical parameters and codebooks after the step
    assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["embed_b"], r[1]["embed_b"])
    assert torch.equal(r[0]["grads"], r[1]["grads"])                 # arena holds the SUM over ranks on every rank
    # serial reference: one process, 2B clips
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4)
    img, gt = make_batch(100, 2 * B, T, H, W)
    recon, latent, _ = tr.step(torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda())
    g_serial = eng.flat_grads.cpu()
    g_ddp = r[0]["grads"] / 2                                         # DDP averages
    assert (g_ddp - g_serial).abs().max().item() <= 2e-3 * g_serial.abs().max().item()
    np.testing.assert_allclose((r[0]["recon"] + r[1]["recon"]).item() / 2, recon.item(), rtol=1e-4)
    np.testing.assert_allclose((r[0]["latent"] + r[1]["latent"]).item() / 2, latent.item(), rtol=1e-4)
    assert (r[0]["embed_b"] - eng.buffers["quantize_b.embed"].cpu()).abs().max().item() <= \
        2e-3 * eng.buffers["quantize_b.embed"].abs().max().item()
    dp = (r[0]["params"] - eng.flat_params.cpu()).abs().max().item()
    assert dp <= 1e-4, dp                                             # one Adam step of lr 3e-4
