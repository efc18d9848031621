#!/usr/bin/env python3
"""Diagnostic (FO_STAMP_H64 build, tools/stamp_h64.sh): where a tile of conv_halo64_bf16_kernel goes, in core clocks, for workgroup 0's wave 0:
issue (next patch's DMAs, the row copy, mask loads), the 36 read -> MFMA half-steps, the epilogue's instruction stream, the wait for the next patch
+ barrier.     python tools/stamp_h64.py [conv1_2|conv2_1] [fwd|dgrad]"""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
os.environ["FACEOFF_HIP_LIB"] = os.path.join(here, "_libfaceoff_stamp_h64.so")
sys.path.insert(0, os.path.dirname(here))
from faceoff_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FRAMES", "160"))
bf = torch.bfloat16
name = sys.argv[1] if len(sys.argv) > 1 else "conv1_2"
kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
H, ci, co = {"conv1_2": (256, 64, 64), "conv2_1": (128, 64, 128)}[name]
cin, cout = (co, ci) if kind == "dgrad" else (ci, co)
x = (torch.randn((N, H, H, cin), device=dev) * 0.5).to(bf)
wp = ops.pack_conv_bf16(torch.randn((cout, cin, 3, 3), device=dev) * 0.05)
out = torch.empty((N, H, H, cout), device=dev, dtype=bf)
b = torch.randn(cout, device=dev)
mask = torch.randn((N, H, H, cout), device=dev).clamp_min(0).to(bf) if kind == "dgrad" else None


def fn():
    ops.conv_bf16(x, wp, b if kind == "fwd" else None, out, cin=cin, cout=cout, flags=ops.FO_OUT_RELU if kind == "fwd" else 0, mask=mask)


for _ in range(100):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    fn()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.fo_debug_read_h64_stamps.argtypes = [C.c_void_p, C.c_int]
lib.fo_debug_read_h64_stamps(buf, 16)
sI, sM, sE, sW, n, cyc, rt = (int(v) for v in buf[:7])
gf = 2.0 * N * H * H * cout * 9 * cin / 1e9
print(f"{name} {kind}: {ms:.3f} ms ({gf / ms:.0f} TFLOP/s, stamped build); {n} tiles by workgroup 0: issue {sI / n:.0f}, MFMA loop {sM / n:.0f}, epilogue {sE / n:.0f}, "
      f"wait + barrier {sW / n:.0f} = {(sI + sM + sE + sW) / n:.0f} clocks per tile ({100.0 * (sI + sM + sE + sW) / cyc:.0f} % of {cyc}); in-kernel clock {cyc / rt * 0.1:.3f} GHz")
