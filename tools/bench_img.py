#!/usr/bin/env python3
"""The 8 -> 64 channel image layer at C2 size (160 frames of 256x256): forward (bias + ReLU) and masked data-gradient form.
    python tools/bench_img.py        (FACEOFF_NO_IMG_KERNEL=1: the tiled kernel)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops
dev = torch.device("cuda:0")
N, H = 160, 256
x = torch.randn((N, H, H, 8), device=dev)
w = torch.randn((64, 6, 4, 4), device=dev) * 0.1
b = torch.randn(64, device=dev)
wp = ops.pack_conv(w)
out = torch.empty((N, H // 2, H // 2, 64), device=dev)
mask = torch.randn_like(out)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
f = lambda: ops.conv_igemm(x, wp, b, out, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=8, cout=64, flags=ops.FO_OUT_RELU)
d = lambda: ops.conv_igemm(x, wp, None, out, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=8, cout=64, mask=mask)
print(f"forward {t(f):.3f} ms   masked {t(d):.3f} ms")
