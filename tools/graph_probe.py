#!/usr/bin/env python3
"""Does one training step capture into a hipGraph (torch.cuda.CUDAGraph) and replay?  python tools/graph_probe.py [clips] [size]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd.trainer import FaceOffTrainer
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = 5
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
tr = FaceOffTrainer(eng)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((B * T, 6, H, H), device=dev, generator=g) * 2 - 1
gt = torch.rand((B * T, 3, H, H), device=dev, generator=g) * 2 - 1
for _ in range(3):
    out = tr.step(img, gt, T=T)
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"eager: {timeit(lambda: tr.step(img, gt, T=T)):.3f} ms/step")
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = tr.step(img, gt, T=T)
torch.cuda.synchronize()
print(f"graph replay: {timeit(graph.replay):.3f} ms/step; losses {[o.item() for o in out]}")
