#!/usr/bin/env python3
"""Do an HBM-bound elementwise kernel (few VGPRs) and the matrix-bound wino_gemm run side by side?  python tools/coexist_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops, _lib
dev = torch.device("cuda:0")
N, P, ci = 160, 36, 128
V = torch.randn(P * N * 256 * ci, device=dev); M = torch.empty_like(V)
U = torch.randn(P * 128 * 3 * 128, device=dev) * 0.05
a = torch.randn((160, 64, 64, 128), device=dev); b = torch.randn_like(a); c = torch.empty_like(a)
x = torch.randn((160, 64, 64, 128), device=dev); Vx = torch.empty(P * N * 256 * ci, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(): _lib.call("fo_wino_gemm", ops._ptr(V), ops._ptr(U), ops._ptr(M), P, N, 5, 256, ci, 128, 3, ops._stream())
def addk():
    for _ in range(3): ops.add(a, b, c)
def xform(): _lib.call("fo_wino_input", ops._ptr(x), 128, ops._ptr(Vx), N, 64, 64, 128, 4, ops._stream())
def t(fn_list, reps=10):
    for _ in range(2):
        for s, f in fn_list:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for s, f in fn_list:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("gemm alone %.3f ms" % t([(s1, gemm)]))
print("3 x add (1 GB each) alone %.3f ms" % t([(s2, addk)]))
print("gemm || 3 x add %.3f ms" % t([(s1, gemm), (s2, addk)]))
print("wino_input alone %.3f ms" % t([(s2, xform)]))
print("gemm || wino_input %.3f ms" % t([(s1, gemm), (s2, xform)]))
