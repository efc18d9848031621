#!/usr/bin/env python3
"""bf16 conv microbenchmarks at the VGG-16 / LPIPS shapes of BASELINE config 3 (160 frames of 256x256):
    python tools/bench_bf16.py [filter]
HIP-event timing (10 reps after 2 warm-ups), nominal TFLOP/s (2*M*Cout*9*Cin)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FRAMES", "160"))
bf = torch.bfloat16


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    shapes = [("conv1_1", 256, 8, 64), ("conv1_2", 256, 64, 64), ("conv2_1", 128, 64, 128), ("conv2_2", 128, 128, 128),
              ("conv3_1", 64, 128, 256), ("conv3_2", 64, 256, 256), ("conv4_1", 32, 256, 512), ("conv4_2", 32, 512, 512),
              ("conv5_x", 16, 512, 512)]
    tot = 0.0
    for name, H, ci, co in shapes:
        for kind in ("fwd", "dgrad"):
            if flt and flt not in f"{name} {kind}":
                continue
            if kind == "dgrad" and ci == 8:
                cin, cout = co, 3
            elif kind == "dgrad":
                cin, cout = co, ci
            else:
                cin, cout = ci, co
            x = (torch.randn((N, H, H, cin), device=dev) * 0.5).to(bf)
            creal = 3 if cin == 8 else cin
            w = torch.randn((cout, creal, 3, 3), device=dev) * 0.05
            if cin == 8:
                wpad = torch.zeros((cout, 8, 3, 3), device=dev)
                wpad[:, :3] = w
                wp = ops.pack_conv_bf16(wpad, taps_pad=16)
            else:
                wp = ops.pack_conv_bf16(w)
            out = torch.empty((N, H, H, (cout + 7) // 8 * 8), device=dev, dtype=bf)
            b = torch.randn(cout, device=dev)
            mask = (torch.randn((N, H, H, cout), device=dev)).clamp_min(0).to(bf) if kind == "dgrad" and cout >= 64 else None
            fn = lambda: ops.conv_bf16(x, wp, b if kind == "fwd" else None, out, cin=cin, cout=cout,
                                       flags=ops.FO_OUT_RELU if kind == "fwd" else 0, mask=mask)
            ms = timeit(fn)
            fl = 2.0 * N * H * H * cout * 9 * creal
            tot += ms
            print(f"{name:8s} {kind:5s} {cin:4d}->{cout:4d} @{H:3d}^2  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s nominal", flush=True)
            del x, out, mask
    print(f"sum {tot:.2f} ms")


if __name__ == "__main__":
    main()
