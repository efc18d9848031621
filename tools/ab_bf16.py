#!/usr/bin/env python3
"""Interleaved A/B of the bf16 conv variants at the C3 shapes, in ONE process on one device (boxes differ by > 10 % in the
clock they hold under an MFMA-dense load):
    python tools/ab_bf16.py "<layer filter>" VAR1 VAR2 ...      VAR = name or name:ENV=1,ENV2=1   (name "base" = no env),
    e.g.  python tools/ab_bf16.py conv4_2 base small:FACEOFF_BF16_SMALL_TILES=1
Each round runs every variant once (HIP events, 5 launches each); prints the median and the minimum over the rounds."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FRAMES", "160"))
ROUNDS = int(os.environ.get("ROUNDS", "9"))
bf = torch.bfloat16
ENVS = ("FACEOFF_BF16_SMALL_TILES", "FACEOFF_BF16_BIG_TILES", "FACEOFF_BF16_NO_DMA", "FACEOFF_BF16_TILE512", "FACEOFF_BF16_NO_PPH", "FACEOFF_BF16_NO_HALO", "FACEOFF_BF16_PPH_PERSIST", "FACEOFF_H64_NO_LINES")


def main():
    flt = sys.argv[1]
    variants = []
    for v in sys.argv[2:]:
        name, _, envs = v.partition(":")
        variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
    shapes = [("conv1_2", 256, 64, 64), ("conv2_1", 128, 64, 128), ("conv2_2", 128, 128, 128), ("conv3_1", 64, 128, 256),
              ("conv3_2", 64, 256, 256), ("conv4_1", 32, 256, 512), ("conv4_2", 32, 512, 512), ("conv5_x", 16, 512, 512),
              ("vq128_64", 64, 128, 128), ("vq128_32", 32, 128, 128)]          # the bf16 VQ-VAE's 128-channel 3x3 layers at the two latent sizes
    for name, H, ci, co in shapes:
        for kind in ("fwd", "dgrad"):
            if flt not in f"{name} {kind}":
                continue
            cin, cout = (co, ci) if kind == "dgrad" else (ci, co)
            x = (torch.randn((N, H, H, cin), device=dev) * 0.5).to(bf)
            wp = ops.pack_conv_bf16(torch.randn((cout, cin, 3, 3), device=dev) * 0.05)
            out = torch.empty((N, H, H, cout), device=dev, dtype=bf)
            b = torch.randn(cout, device=dev)
            mask = torch.randn((N, H, H, cout), device=dev).clamp_min(0).to(bf) if kind == "dgrad" else None

            def fn():
                ops.conv_bf16(x, wp, b if kind == "fwd" else None, out, cin=cin, cout=cout,
                              flags=ops.FO_OUT_RELU if kind == "fwd" else 0, mask=mask)
            times = {v[0]: [] for v in variants}
            for r in range(ROUNDS + 1):
                for vname, env in variants:
                    for e in ENVS:
                        os.environ.pop(e, None)
                    os.environ.update(env)
                    fn()
                    s, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(5):
                        fn()
                    e2.record()
                    torch.cuda.synchronize()
                    if r:
                        times[vname].append(s.elapsed_time(e2) / 5)
            gf = 2.0 * N * H * H * cout * 9 * cin / 1e9
            print(f"{name} {kind:5s} " + "  ".join(f"{v}: med {statistics.median(t):.3f} ms ({gf / statistics.median(t):.0f} TF) min {min(t):.3f}"
                                                  for v, t in times.items()), flush=True)


main()
