#!/usr/bin/env python3
"""Winograd vs direct Conv3d at the C2 shapes:  python tools/bench_wino.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops
dev = torch.device("cuda:0")
N, T = 160, 5

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

for H in (64, 32):
    x = torch.randn((N, H, H, 128), device=dev)
    w = torch.randn((128, 128, 3, 3, 3), device=dev) * 0.02
    b = torch.randn(128, device=dev)
    out = torch.empty_like(x)
    wp = ops.pack_conv(w); U = ops.wino_filter(w)
    mask = torch.randn_like(x).clamp_min(0)
    t_d = timeit(lambda: ops.conv_igemm(x, wp, b, out, T=T, k=(3, 3, 3), pad=(1, 1, 1), cin=128, cout=128, flags=ops.FO_OUT_RELU))
    t_w = timeit(lambda: ops.conv3d_winograd(x, U, b, out, T=T, cin=128, cout=128, flags=ops.FO_OUT_RELU))
    t_wm = timeit(lambda: ops.conv3d_winograd(x, U, None, out, T=T, cin=128, cout=128, mask=mask))
    prof = ops.KernelProfiler(detail=True); ops.PROFILER = prof
    ops.conv3d_winograd(x, U, b, out, T=T, cin=128, cout=128, flags=ops.FO_OUT_RELU)
    ops.PROFILER = None
    g = prof.summary()
    gemm = sum(v["total_ms"] for v in g.values())
    print(f"{H}^2: direct {t_d:.3f} ms   winograd {t_w:.3f} ms (masked dgrad form {t_wm:.3f})   of which GEMMs {gemm:.3f} ms "
          + ", ".join(f"{v['tflops']:.0f} TF" for v in g.values()))

print("wgrad:")
for H in (64, 32):
    x = torch.randn((N, H, H, 128), device=dev)
    g = torch.randn((N, H, H, 128), device=dev)
    dw = torch.empty((128, 128, 3, 3, 3), device=dev); db = torch.empty(128, device=dev)
    t_d = timeit(lambda: ops.conv_wgrad(g, x, dw, db, T=T, k=(3, 3, 3), pad=(1, 1, 1), a_real=128, b_real=128))
    t_w = timeit(lambda: ops.conv3d_wgrad_winograd(g, x, dw, db, T=T, a_real=128, b_real=128))
    prof = ops.KernelProfiler(detail=True); ops.PROFILER = prof
    ops.conv3d_wgrad_winograd(g, x, dw, db, T=T, a_real=128, b_real=128)
    ops.PROFILER = None
    gm = prof.summary()
    print(f"{H}^2: direct {t_d:.3f} ms   winograd {t_w:.3f} ms   of which GEMM " + ", ".join(f"{v['total_ms']:.3f} ms {v['tflops']:.0f} TF" for v in gm.values()))

print("2-D 3x3 128->128 @64^2:")
x = torch.randn((N, 64, 64, 128), device=dev); g = torch.randn((N, 64, 64, 128), device=dev)
w = torch.randn((128, 128, 3, 3), device=dev) * 0.03; b = torch.randn(128, device=dev); out = torch.empty_like(x)
wp = ops.pack_conv(w); dw = torch.empty_like(w); db = torch.empty(128, device=dev)
t_d = timeit(lambda: ops.conv_igemm(x, wp, b, out, k=(1, 3, 3), pad=(0, 1, 1), cin=128, cout=128))
t_dw = timeit(lambda: ops.conv_wgrad(g, x, dw, db, k=(1, 3, 3), pad=(0, 1, 1), a_real=128, b_real=128))
for m in (2, 4):
    U = ops.wino_filter(w, m=m)
    t_w = timeit(lambda: ops.conv3d_winograd(x, U, b, out, T=1, cin=128, cout=128, m=m, kd=1))
    V = ops.conv3d_winograd(x, U, b, out, T=1, cin=128, cout=128, m=m, kd=1, keep_v=True)
    t_ww = timeit(lambda: ops.conv3d_wgrad_winograd(g, x, dw, db, T=1, a_real=128, b_real=128, V=V, m=m, kd=1))
    print(f"  F{m}: fwd direct {t_d:.3f} ms  winograd {t_w:.3f} ms ;  wgrad direct {t_dw:.3f} ms  winograd (V kept) {t_ww:.3f} ms")
