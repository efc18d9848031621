#!/usr/bin/env python3
"""Soak: N training steps at C2 size, reporting step time and allocator state every 50 (leaks / drift):  python tools/soak.py [N]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd.trainer import FaceOffTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
tr = FaceOffTrainer(eng)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((160, 6, 256, 256), device=dev, generator=g) * 2 - 1
gt = torch.rand((160, 3, 256, 256), device=dev, generator=g) * 2 - 1
t0 = time.perf_counter()
for i in range(1, n + 1):
    recon, latent, _ = tr.step(img, gt, T=5)
    if i % 50 == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3
        print(f"step {i}: {dt:.2f} ms/step  recon {recon.item():.5f} latent {latent.item():.5f}  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        t0 = time.perf_counter()
