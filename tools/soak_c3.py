#!/usr/bin/env python3
"""Soak of config 3 as timed (bf16 VQ-VAE + bf16 LPIPS in one FaceOffTrainer.step): N steps, reporting step time, losses and allocator state every 50;
the recon loss must fall and stay finite.     python tools/soak_c3.py [N]"""
import os, sys, time, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.loss import VQLPIPS
from faceoff_amd.synth import make_state_dict, make_vgg_lpips_state
from faceoff_amd.trainer import FaceOffTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev, dtype="bf16")
tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(make_vgg_lpips_state(7), dtype="bf16").to(dev))
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((160, 6, 256, 256), device=dev, generator=g) * 2 - 1
gt = torch.rand((160, 3, 256, 256), device=dev, generator=g) * 2 - 1
t0 = time.perf_counter()
first = None
for i in range(1, n + 1):
    recon, latent, perc = tr.step(img, gt, T=5)
    if i % 50 == 0 or i == 1:
        torch.cuda.synchronize()
        r, l, p = recon.item(), latent.item(), perc.item()
        assert all(math.isfinite(v) for v in (r, l, p)), (r, l, p)
        first = first or r
        dt = (time.perf_counter() - t0) / (50 if i > 1 else 1) * 1e3
        print(f"step {i}: {dt:.2f} ms/step  recon {r:.5f} latent {l:.5f} perceptual {p:.5f}  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        t0 = time.perf_counter()
assert r < first, "recon loss did not fall"
print("soak ok")
