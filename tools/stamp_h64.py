#!/usr/bin/env python3
"""Where a tile of conv_halo64_bf16_kernel goes (in-kernel s_memtime stamps of workgroup 0, wave 0; a -DFO_STAMP_H64 build of conv_bf16.hip):
    FACEOFF_HIP_LIB=faceoff_amd/csrc/variants/lib_stamp.so python tools/stamp_h64.py
conv1_2 (64 -> 64 at 160 x 256 x 256) as plain forward, forward + pool + codes + plane, and the masked data gradient (bit plane)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import _lib, ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
N, H = int(os.environ.get("FRAMES", "160")), 256
x = (torch.randn((N, H, H, 64), device=dev) * 0.5).to(bf)
wp = ops.pack_conv_bf16(torch.randn((64, 64, 3, 3), device=dev) * 0.05)
b = torch.randn(64, device=dev)
out = torch.empty((N, H, H, 64), device=dev, dtype=bf)
pooled = torch.empty((N, H // 2, H // 2, 64), device=dev, dtype=bf)
pidx = torch.empty((N, H // 2, H // 2, 16), device=dev, dtype=torch.uint8)
pbits = torch.empty((N, H // 2, H // 2, 8), device=dev, dtype=torch.uint8)
mbits = torch.randint(0, 256, (N, H, H, 8), device=dev, dtype=torch.uint8)
lib = _lib.load()
lib.fo_debug_read_h64_stamps.argtypes = [C.c_void_p, C.c_int]
cases = {
    "forward": lambda: ops.conv_bf16(x, wp, b, out, cin=64, cout=64, flags=ops.FO_OUT_RELU),
    "forward + pool + codes + plane": lambda: ops.conv_bf16(x, wp, b, out, cin=64, cout=64, flags=ops.FO_OUT_RELU, pooled=pooled, pool_idx=pidx, pooled_bits=pbits),
    "masked data gradient (bit plane)": lambda: ops.conv_bf16(x, wp, None, out, cin=64, cout=64, mask_bits=mbits),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record()
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    assert lib.fo_debug_read_h64_stamps(st, 16) == 0
    issue, mfma, epi, wait, nT, kclk, rclk = [int(v) for v in st[:7]]
    ms = s.elapsed_time(e)
    ghz = kclk / (rclk / 100e6) / 1e9 if rclk else 0.0          # s_memrealtime: 100 MHz
    print(f"{name}: {ms:.3f} ms ; wave 0 of workgroup 0: {nT} tiles, kernel {kclk} core clocks = {rclk / 100e6 * 1e3:.3f} ms at {ghz:.2f} GHz; per tile "
          f"issue {issue / nT:.0f}  MFMA loop {mfma / nT:.0f}  epilogue {epi / nT:.0f}  wait+barrier {wait / nT:.0f}  = {(issue + mfma + epi + wait) / nT:.0f} clocks")
