#!/usr/bin/env python3
"""Robustness sweep: one training step at several batch geometries on the default engine (Winograd forms, image-layer kernels,
deferred filter gradients) against the same step on the direct kernels: losses and all 70 gradients must agree.
    python tools/shape_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd import ops
dev = torch.device("cuda:0")
sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
bad = 0
for (B, T, H, W) in [(8, 5, 256, 256), (32, 5, 128, 128), (3, 3, 256, 192), (1, 5, 64, 96), (16, 2, 256, 256), (5, 1, 200, 200), (64, 5, 64, 64)]:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + H)
    img = torch.rand((B * T, 6, H, W), device=dev, generator=g) * 2 - 1
    gt = torch.rand((B * T, 3, H, W), device=dev, generator=g) * 2 - 1
    res = []
    for direct in (False, True):
        junk = torch.full((1 << 30,), float("nan"), device=dev)       # poison the allocator's free blocks (4 GiB of NaN)
        del junk
        eng = VQVAEEngine(sd, dev)
        if direct:
            eng.winograd = False
            eng.defer_wgrad = False
            ops.W42 = False
        else:
            ops.W42 = True
        recon, latent = eng.loss_and_backward(img, gt, T=T)[:2]
        torch.cuda.synchronize()
        res.append((recon.item(), latent.item(), {k: v.clone() for k, v in eng.grads.items()}))
    (r0, l0, g0), (r1, l1, g1) = res
    worst = max(((g0[k] - g1[k]).abs().max().item() / (g1[k].abs().max().item() + 1e-30), k) for k in g1)
    ok = abs(r0 - r1) <= 1e-4 * abs(r1) and abs(l0 - l1) <= 1e-3 * abs(l1) and worst[0] < 5e-2 and all(torch.isfinite(v).all() for v in g0.values())
    bad += not ok
    print(f"B={B} T={T} {H}x{W}: recon {r0:.6f}/{r1:.6f} latent {l0:.6f}/{l1:.6f} worst grad rel diff {worst[0]:.1e} ({worst[1]}) {'ok' if ok else 'MISMATCH'}", flush=True)
print("SWEEP", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
