#!/usr/bin/env python3
"""Diagnostic (FO_STAMP build): per-K-step cycle anatomy of workgroup 0 / wave 0 of a 128x128 k3 wgrad."""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["FACEOFF_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
N, H = 160, 64
x = torch.randn(N, H, H, 128, device=dev)
g = torch.randn(N, H, H, 128, device=dev)
dw = torch.empty(128, 128, 3, 3, device=dev)
for _ in range(3):
    ops.conv_wgrad(g, x, dw, None, k=(1, 3, 3), pad=(0, 1, 1), a_real=128, b_real=128)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 4096)()
lib.fo_debug_read_wstamps.argtypes = [C.c_void_p, C.c_int]
lib.fo_debug_read_wstamps(buf, 4096)
st = np.array(buf[:], dtype=np.uint64).astype(np.int64)
ns = 150
st = st[:8 * ns].reshape(ns, 8)
seg = np.diff(st[:, :7], axis=1)
gap = st[1:, 0] - st[:-1, 6]
names = ["top (next_valid+walk)", "g0 (+loads)", "g1", "g2", "g3 (+lds writes)", "tail"]
for lo, hi, tag in ((0, 10, "first 10 steps"), (40, 150, "steps 40..149")):
    print(tag)
    for i, nm in enumerate(names):
        print("   %-22s median %6.0f  min %6d  max %6d" % (nm, np.median(seg[lo:hi, i]), seg[lo:hi, i].min(), seg[lo:hi, i].max()))
    gg = gap[lo:hi - 1]
    print("   %-22s median %6.0f  min %6d  max %6d" % ("barrier gap", np.median(gg), gg.min(), gg.max()))
    print("   per step total median %.0f (ideal 8192 with a partner wave, 4096 alone)" % np.median(st[lo + 1:hi, 0] - st[lo:hi - 1, 0]))
