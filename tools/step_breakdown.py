#!/usr/bin/env python3
"""Per-launch-shape breakdown of one C2 training step (HIP events):  python tools/step_breakdown.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402
from faceoff_amd.engine import VQVAEEngine  # noqa: E402
from faceoff_amd.synth import make_state_dict  # noqa: E402
from faceoff_amd.trainer import FaceOffTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, T, H = 32, 5, 256
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev, dtype="bf16" if "--bf16" in sys.argv else "fp32")
vqlpips = None
if "--lpips" in sys.argv:
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.synth import make_vgg_lpips_state
    vqlpips = VQLPIPS(make_vgg_lpips_state(7), dtype="bf16").to(dev)
tr = FaceOffTrainer(eng, vqlpips=vqlpips)
if "--overlap" not in sys.argv:          # default: every kernel alone on the GPU (side streams folded into the main one)
    eng.set_stream_overlap(False)
gen = torch.Generator(device=dev).manual_seed(1234)
img = torch.rand((B * T, 6, H, H), device=dev, generator=gen) * 2 - 1
gt = torch.rand((B * T, 3, H, H), device=dev, generator=gen) * 2 - 1
for _ in range(2):
    tr.step(img, gt, T=T)
torch.cuda.synchronize()
steps = 3
prof = ops.KernelProfiler(detail=True)
ops.PROFILER = prof
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(img, gt, T=T)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
ops.PROFILER = None
summ = prof.summary()
tot = sum(v["total_ms"] for v in summ.values()) / steps
print(f"step {dt:.2f} ms; conv launches {tot:.2f} ms; other {dt - tot:.2f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{v['total_ms'] / steps:8.3f} ms/step  x{v['launches'] // steps:<3d} {v['avg_ms']:8.3f} ms  {v['tflops']:6.1f} TF  {k}")
