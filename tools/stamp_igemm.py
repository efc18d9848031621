#!/usr/bin/env python3
"""Diagnostic (FO_STAMP build): per-K-step cycle counts of workgroup 0 / wave 0 of the conv3d_b forward."""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["FACEOFF_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
N, T, H = 160, 5, 64
x = torch.randn(N, H, H, 128, device=dev)
w = torch.randn(128, 128, 3, 3, 3, device=dev) * 0.05
wp = ops.pack_conv(w)
out = torch.empty_like(x)
for _ in range(3):
    ops.conv_igemm(x, wp, None, out, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=128, cout=128)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 4096)()
lib.fo_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
lib.fo_debug_read_stamps(buf, 4096)
st = np.array(buf[:], dtype=np.uint64).astype(np.int64)
ns = 72
st = st[:8 * ns].reshape(ns, 8)
seg = np.diff(st[:, :6], axis=1)          # g0, g1, g2, g3, pre-barrier tail
gap = st[1:, 0] - st[:-1, 5]              # barrier + loop top
names = ["g0(+loads)", "g1", "g2", "g3(+lds writes)", "tail"]
for lo, hi, tag in ((0, 12, "alone (first 12 steps)"), (24, 72, "with partner (steps 24..71)")):
    print(tag)
    for i, nm in enumerate(names):
        print("   %-16s median %6.0f  min %6d  max %6d" % (nm, np.median(seg[lo:hi, i]), seg[lo:hi, i].min(), seg[lo:hi, i].max()))
    g = gap[lo:hi - 1]
    print("   %-16s median %6.0f  min %6d  max %6d" % ("barrier gap", np.median(g), g.min(), g.max()))
    print("   per step total median %.0f" % np.median(st[lo + 1:hi, 0] - st[lo:hi - 1, 0]))
