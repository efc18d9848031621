#!/usr/bin/env python3
"""Filter gradient of the 8-channel image layer at C2 size:  python tools/bench_wimg.py   (FACEOFF_NO_IMG_KERNEL=1: tiled kernel)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops
dev = torch.device("cuda:0")
N, H = 160, 256
x = torch.randn((N, H, H, 8), device=dev)
g = torch.randn((N, H // 2, H // 2, 64), device=dev)
dw, db = torch.empty((64, 6, 4, 4), device=dev), torch.empty(64, device=dev)
f = lambda: ops.conv_wgrad(g, x, dw, db, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=64, b_real=6)
for _ in range(3): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
print(f"wgrad {s.elapsed_time(e) / 10:.3f} ms")
