#!/usr/bin/env python3
"""Microbenchmark of the bf16 filter-gradient kernel at the C2 shapes:  python tools/bench_wgrad_bf16.py [filter] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402

flt = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, T = 160, 5
CASES = [  # name, ca, cb, k, stride, pad, Hm, Wm, in_relu
    ("conv3d64", 128, 128, (3, 3, 3), 1, (1, 1, 1), 64, 64, False),
    ("conv3d32", 128, 128, (3, 3, 3), 1, (1, 1, 1), 32, 32, False),
    ("c3x3_128", 128, 128, (1, 3, 3), 1, (0, 1, 1), 64, 64, False),
    ("res3x3", 32, 128, (1, 3, 3), 1, (0, 1, 1), 64, 64, True),
    ("res1x1", 128, 32, (1, 1, 1), 1, (0, 0, 0), 64, 64, False),
    ("k4s2_128x64", 128, 64, (1, 4, 4), 2, (0, 1, 1), 64, 64, False),
    ("img", 64, 8, (1, 4, 4), 2, (0, 1, 1), 128, 128, False),
]
for name, ca, cb, k, s, pad, Hm, Wm, relu in CASES:
    if flt and flt not in name:
        continue
    Hq, Wq = (Hm * 2, Wm * 2) if s == 2 else (Hm, Wm)
    P = torch.randn((N, Hm, Wm, ca), device="cuda").to(torch.bfloat16)
    Q = torch.randn((N, Hq, Wq, cb), device="cuda").to(torch.bfloat16)
    dw = torch.empty((ca, cb, k[0] * k[1] * k[2]), device="cuda")
    db = torch.empty(ca, device="cuda")
    for _ in range(3):
        ops.conv_wgrad_bf16(P, Q, dw, db, T=T if k[0] > 1 else 1, k=k, stride=s, pad=pad, a_real=ca, b_real=cb, in_relu=relu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.conv_wgrad_bf16(P, Q, dw, db, T=T if k[0] > 1 else 1, k=k, stride=s, pad=pad, a_real=ca, b_real=cb, in_relu=relu)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * N * Hm * Wm * ca * cb * k[0] * k[1] * k[2] * (ops.temporal_share(T) if k[0] > 1 else 1.0)
    print(f"{name:14s} {dt * 1e3:8.3f} ms  {fl / dt / 1e12:7.1f} TFLOP/s")
