#!/usr/bin/env python3
"""One Winograd Conv3d at the C2 size, whole batch vs the batch-halves pipeline:  python tools/split_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops
dev = torch.device("cuda:0")
N, T = 160, 5
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for H in (64, 32):
    x = torch.randn((N, H, H, 128), device=dev); g = torch.randn_like(x)
    w = torch.randn((128, 128, 3, 3, 3), device=dev) * 0.02; b = torch.randn(128, device=dev)
    out = torch.empty_like(x); U = ops.wino_filter(w, m=4)
    dw, db = torch.empty_like(w), torch.empty(128, device=dev)
    for split in (False, True):
        ops.WINO_SPLIT = split
        tf = t(lambda: ops.conv3d_winograd(x, U, b, out, T=T, cin=128, cout=128, flags=ops.FO_OUT_RELU, m=4, kd=3))
        V = ops.conv3d_winograd(x, U, b, out, T=T, cin=128, cout=128, flags=ops.FO_OUT_RELU, m=4, kd=3, keep_v=True)
        tw = t(lambda: ops.conv3d_wgrad_winograd(g, x, dw, db, T=T, a_real=128, b_real=128, V=V, m=4, kd=3))
        print(f"{H}^2 split={split}: conv {tf:.3f} ms   wgrad (V kept) {tw:.3f} ms")
