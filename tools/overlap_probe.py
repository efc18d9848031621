#!/usr/bin/env python3
"""Upper bound of micro-batch pipelining: one 160-frame step vs two independent 80-frame steps enqueued on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd.trainer import FaceOffTrainer
dev = torch.device("cuda:0")
T, H = 5, 256
gen = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((160, 6, H, H), device=dev, generator=gen) * 2 - 1
gt = torch.rand((160, 3, H, H), device=dev, generator=gen) * 2 - 1
sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)

def timeit(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

tr = FaceOffTrainer(VQVAEEngine(sd, dev))
print("one step, 160 frames: %.2f ms" % timeit(lambda: tr.step(img, gt, T=T)))
for parts in (2, 4):
    n = 160 // parts
    trs = [FaceOffTrainer(VQVAEEngine(sd, dev)) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def both():
        for i, (t, s) in enumerate(zip(trs, streams)):
            with torch.cuda.stream(s):
                t.step(img[i * n:(i + 1) * n], gt[i * n:(i + 1) * n], T=T)
    print("%d concurrent steps of %d frames: %.2f ms" % (parts, n, timeit(both)))
    t1 = trs[0]
    print("   (one %d-frame step alone: %.2f ms)" % (n, timeit(lambda: t1.step(img[:n], gt[:n], T=T))))
