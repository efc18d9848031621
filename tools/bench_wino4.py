#!/usr/bin/env python3
"""Piece-by-piece timing of the Winograd F(4x4,3x3) path at the C2 shapes (HIP events, each kernel alone):
input transform, banked GEMM, output transform, output-gradient transform, banked filter-gradient GEMM.
    python tools/bench_wino4.py [reps]
Prints ms, TFLOP/s (GEMMs, executed FLOP) or TB/s (transforms, algorithmic bytes)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
N, T, m = 160, 5, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for H, kd in ((64, 3), (32, 3), (64, 1)):
    Tt = T if kd == 3 else 1
    ci = co = 128
    Ht = Wt = H // m
    P = 36
    x = torch.randn((N, H, H, ci), device=dev)
    g = torch.randn((N, H, H, co), device=dev)
    w = torch.randn((co, ci, 3, 3, 3) if kd == 3 else (co, ci, 3, 3), device=dev) * 0.02
    b = torch.randn(co, device=dev)
    out = torch.empty_like(x)
    U = ops.wino_filter(w, m=m)
    plane = N * Ht * Wt * ci
    V = torch.empty(P * plane, device=dev)
    M = torch.empty(P * plane, device=dev)
    st = ops._stream
    t_in = timeit(lambda: _lib.call("fo_wino_input", ops._ptr(x), ci, ops._ptr(V), N, H, H, ci, m, st()))
    d = ops._desc(N=P * N, T=Tt, Hin=Ht, Win=Wt, Hm=Ht, Wm=Wt, Hout=Ht, Wout=Wt, Cin=ci, Cout=co, KD=kd, KH=1, KW=1, stride=1,
                  padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=ci, ldOut=co, ldMask=0, ldAdd=0, flags=0)
    t_g_old = timeit(lambda: _lib.call("fo_conv_igemm_banked", C.byref(d), ops._ptr(V), ops._ptr(U), ops._ptr(M), N, st()))
    t_g_new = None
    if hasattr(_lib.load(), "fo_wino_gemm"):
        t_g_new = timeit(lambda: _lib.call("fo_wino_gemm", ops._ptr(V), ops._ptr(U), ops._ptr(M), P, N, Tt, Ht * Wt, ci, co, kd, st()))
    t_out = timeit(lambda: _lib.call("fo_wino_output", ops._ptr(M), ops._ptr(b), None, 0, None, 0, ops._ptr(out), co, N, H, H, co,
                                     ops.FO_BIAS | ops.FO_OUT_RELU, m, st()))
    t_go = timeit(lambda: _lib.call("fo_wino_gradout", ops._ptr(g), co, ops._ptr(M), N, H, H, co, m, st()))
    dw, db = torch.empty_like(w), torch.empty(co, device=dev)
    t_wg = timeit(lambda: ops.conv3d_wgrad_winograd(g, x, dw, None, T=Tt, a_real=co, b_real=ci, V=V, m=m, kd=kd))
    flop = 2.0 * P * N * Ht * Wt * co * ci * kd * (ops.temporal_share(Tt) if kd == 3 else 1.0)
    xb, pb = x.numel() * 4, P * plane * 4
    line = (f"{H}^2 kd={kd}: input {t_in:.3f} ms ({(xb + pb) / t_in / 1e9:.2f} TB/s)  gemm(banked igemm) {t_g_old:.3f} ms "
            f"({flop / t_g_old / 1e9:.1f} TF)")
    if t_g_new is not None:
        line += f"  gemm(wino_gemm) {t_g_new:.3f} ms ({flop / t_g_new / 1e9:.1f} TF)"
    line += (f"  output {t_out:.3f} ms ({(xb + pb) / t_out / 1e9:.2f} TB/s)  gradout {t_go:.3f} ms ({(xb + pb) / t_go / 1e9:.2f} TB/s)"
             f"  wgrad total (V kept: gradout + GEMM + G^T dU G) {t_wg:.3f} ms ({flop / t_wg / 1e9:.1f} TF)")
    print(line)
