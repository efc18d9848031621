#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its GPU time (is the step launch-bound?):  python tools/host_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd.trainer import FaceOffTrainer
dev = torch.device("cuda:0")
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
tr = FaceOffTrainer(eng)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((160, 6, 256, 256), device=dev, generator=g) * 2 - 1
gt = torch.rand((160, 3, 256, 256), device=dev, generator=g) * 2 - 1
for _ in range(3):
    tr.step(img, gt, T=5)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(5):
    a = time.perf_counter()
    tr.step(img, gt, T=5)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 5
print(f"step {tot*1e3:.1f} ms; host enqueue per step: " + ", ".join(f"{h*1e3:.1f}" for h in host) + " ms")
