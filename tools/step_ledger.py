#!/usr/bin/env python3
"""Every C-ABI launch of one training step in launch order with its own bound (ops.StepLedger: algorithmic bytes / FLOP -> floor) beside the time
it took alone on the GPU -- where a step stands above its attainable floor, launch by launch.
    python tools/step_ledger.py [--bf16] [--lpips] [--gan] [--top N | --by-kernel]     (C2 fp32 by default; --bf16 --lpips = config 3; --gan = config 5;
                                                                                        --by-kernel: one markdown row per kernel symbol, for profiles/)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the constants the bench line prices floors at)
from faceoff_amd import ops  # noqa: E402
from faceoff_amd.engine import VQVAEEngine  # noqa: E402
from faceoff_amd.synth import make_state_dict  # noqa: E402
from faceoff_amd.trainer import FaceOffTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, T, H = 32, 5, 256
gen = torch.Generator(device=dev).manual_seed(1234)
img = torch.rand((B * T, 6, H, H), device=dev, generator=gen) * 2 - 1
gt = torch.rand((B * T, 3, H, H), device=dev, generator=gen) * 2 - 1
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev, dtype="bf16" if "--bf16" in sys.argv else "fp32")
if "--gan" in sys.argv:
    import random
    from faceoff_amd.disc import DiscEngine
    from faceoff_amd.gan_trainer import GANTrainer
    from faceoff_amd.synth import make_disc_state
    gan = GANTrainer(eng, DiscEngine(make_disc_state(1, 3), dev, dims=3, n_frames=15), DiscEngine(make_disc_state(2, 2), dev, dims=2), window=16, rng=random.Random(7))
    gan.overlap_d2 = False
    gan.d3.overlap_scales = gan.d2.overlap_scales = False
    cimg, cgt = img[:30].contiguous(), gt[:30].contiguous()
    step, per = (lambda: gan.step(cimg, cgt)), 2
else:
    vqlpips = None
    if "--lpips" in sys.argv:
        from faceoff_amd.loss import VQLPIPS
        from faceoff_amd.synth import make_vgg_lpips_state
        vqlpips = VQLPIPS(make_vgg_lpips_state(7), dtype="bf16").to(dev)
    tr = FaceOffTrainer(eng, vqlpips=vqlpips)
    step, per = (lambda: tr.step(img, gt, T=T)), 1
eng.set_stream_overlap(False)
for _ in range(4):
    step()
torch.cuda.synchronize()
led = ops.StepLedger().open()
for _ in range(per):
    step()
torch.cuda.synchronize()
led.close()
is16 = lambda k: "bf16" in k
rows = []
for i, c in enumerate(led.calls):
    t_b = c["bytes"] / bench.HBM_BW_ACHIEVABLE * 1e3
    pk = bench.BF16_MFMA_PEAK_TFLOPS if is16(c["symbol"]) else bench.FP32_MFMA_PEAK_TFLOPS
    t_f = c["flops"] / (pk * 1e12 * bench.HELD_CLOCK_GHZ["bf16" if is16(c["symbol"]) else "f32"] / 2.4) * 1e3
    rows.append((i, c["symbol"], c["entry"], c["bytes"] / 1e6, c["flops"] / 1e9, c["ev0"].elapsed_time(c["ev1"]), max(t_b, t_f), "hbm" if t_b >= t_f else "mfma"))
tot_m, tot_f = sum(r[5] for r in rows), sum(r[6] for r in rows)
print(f"{len(rows)} launches; measured (alone) {tot_m:.2f} ms; floor {tot_f:.2f} ms; gap {tot_m - tot_f:.2f} ms")
if "--by-kernel" in sys.argv:
    agg = {}
    for i, sym, ent, mb, gf, ms, fl, b in rows:
        a = agg.setdefault(sym, [0, 0.0, 0.0, 0.0, 0.0, {"hbm": 0, "mfma": 0}])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += mb; a[4] += gf; a[5][b] += 1
    k = 1.0 / per
    print(f"\n| kernel | launches | measured ms (alone) | floor ms | measured / floor | bound | algorithmic MB | algorithmic GFLOP |\n|---|---|---|---|---|---|---|---|")
    for sym, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] * k < 0.02:
            continue
        print(f"| `{sym}` | {a[0] * k:g} | {a[1] * k:.3f} | {a[2] * k:.3f} | {a[1] / max(a[2], 1e-9):.2f} | {'hbm' if a[5]['hbm'] >= a[5]['mfma'] else 'mfma'} | {a[3] * k:.0f} | {a[4] * k:.1f} |")
    rest = [a for a in agg.values() if a[1] * k < 0.02]
    print(f"| ({len(rest)} kernels under 0.02 ms) | {sum(a[0] for a in rest) * k:g} | {sum(a[1] for a in rest) * k:.3f} | {sum(a[2] for a in rest) * k:.3f} | | | | |")
    print(f"| **total** | {len(rows) * k:g} | {tot_m * k:.2f} | {tot_f * k:.2f} | {tot_m / tot_f:.2f} | | | |")
    raise SystemExit(0)
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
order = sorted(rows, key=lambda r: -(r[5] - r[6]))[:top] if top else rows
print(f"{'#':>4} {'measured':>9} {'floor':>8} {'gap':>8} bound {'MB':>9} {'GFLOP':>9}  kernel (entry)")
for i, sym, ent, mb, gf, ms, fl, b in order:
    print(f"{i:4d} {ms:9.4f} {fl:8.4f} {ms - fl:8.4f} {b:5s} {mb:9.1f} {gf:9.1f}  {sym} ({ent})")
