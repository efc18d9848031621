#!/usr/bin/env python3
"""Diagnostic (FO_STAMP_PPH build, tools/stamp_pph.sh): where a phase of conv_bf16_pph_kernel's K loop goes, in core clocks, for workgroup 0's wave 0
(group G0) and wave 4 (G1): R = phase start -> past the first barrier (fragment reads, DMA issue, G1's counted wait, barrier), M = the 32 MFMAs'
issue, W = G0's counted wait + the second barrier.  Also the kernel's in-kernel clock (s_memtime / s_memrealtime x 100 MHz).
    python tools/stamp_pph.py "conv4_2 fwd" [lib suffix]"""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
os.environ["FACEOFF_HIP_LIB"] = os.path.join(here, "_libfaceoff_stamp_pph%s.so" % (sys.argv[2] if len(sys.argv) > 2 else ""))
sys.path.insert(0, os.path.dirname(here))
from faceoff_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FRAMES", "160"))
bf = torch.bfloat16
shapes = {"conv2_2": (128, 128, 128), "conv3_2": (64, 256, 256), "conv4_2": (32, 512, 512), "conv5_x": (16, 512, 512), "vq128_64": (64, 128, 128), "vq128_32": (32, 128, 128)}
name, kind = sys.argv[1].split()
H, ci, co = shapes[name]
cin, cout = (co, ci) if kind == "dgrad" else (ci, co)
x = (torch.randn((N, H, H, cin), device=dev) * 0.5).to(bf)
wp = ops.pack_conv_bf16(torch.randn((cout, cin, 3, 3), device=dev) * 0.05)
out = torch.empty((N, H, H, cout), device=dev, dtype=bf)
b = torch.randn(cout, device=dev)
mask = torch.randn((N, H, H, cout), device=dev).clamp_min(0).to(bf) if kind == "dgrad" else None


def fn():
    ops.conv_bf16(x, wp, b if kind == "fwd" else None, out, cin=cin, cout=cout, flags=ops.FO_OUT_RELU if kind == "fwd" else 0, mask=mask)


for _ in range(200):      # (the clock the chip settles at under this load)
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
buf = (C.c_ulonglong * 64)()
lib.fo_debug_read_pph_stamps.argtypes = [C.c_void_p, C.c_int]
lib.fo_debug_read_pph_stamps(buf, 64)
gf = 2.0 * N * H * H * cout * 9 * cin / 1e9
print(f"{name} {kind}: {ms:.3f} ms ({gf / ms:.0f} TFLOP/s, stamped build)")
for g in (0, 1):
    sR, sM, sW, n, cyc, rt, sSet, sEpi, sWt, ntl = (int(v) for v in buf[g * 32:g * 32 + 10])
    sF = [int(v) for v in buf[g * 32 + 10:g * 32 + 19]]
    if not n:
        continue
    print(f"  G{g}: phases {n}  R {sR / n:.0f}  M {sM / n:.0f}  W {sW / n:.0f}  = {(sR + sM + sW) / n:.0f} clocks per phase;  K loop {100.0 * (sR + sM + sW) / cyc:.1f} % of the workgroup's "
          f"{cyc} clocks;  in-kernel clock {cyc / rt * 0.1:.3f} GHz")
    if ntl:
        print(f"      between tiles ({ntl}): next tile's setup + prologue issue {sSet / ntl:.0f}, epilogue instructions {sEpi / ntl:.0f}, wait for the prologue {sWt / ntl:.0f} clocks;"
              f"  phases 0..8 of a tile: " + " ".join(f"{v / (ntl + 1):.0f}" for v in sF))
