#!/usr/bin/env python3
"""Host-side enqueue time of one GAN iteration vs its GPU time (is config 5 launch-bound?):  python tools/host_time_gan.py"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.disc import DiscEngine
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.gan_trainer import GANTrainer
from faceoff_amd.synth import make_state_dict, make_disc_state
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((30, 6, 256, 256), device=dev, generator=gen) * 2 - 1
gt = torch.rand((30, 3, 256, 256), device=dev, generator=gen) * 2 - 1
tr = GANTrainer(VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev), DiscEngine(make_disc_state(1, 3), dev, dims=3, n_frames=15),
                DiscEngine(make_disc_state(2, 2), dev, dims=2), rng=random.Random(3))
for _ in range(4):
    tr.step(img, gt)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter()
    tr.step(img, gt)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 8
print(f"iteration {tot*1e3:.1f} ms; host enqueue per iteration (G, D alternating): " + ", ".join(f"{h*1e3:.1f}" for h in host) + " ms")
