#!/usr/bin/env python3
"""Per-kernel microbenchmarks at BASELINE C2 shapes (run on the GPU box):  python tools/bench_kernels.py [filter]
Times each launch shape with HIP events (20 reps after 3 warm-ups) and prints nominal TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, T = 160, 5


def timeit(fn, reps=20):
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


def rnd(*shape):
    return torch.randn(shape, device=dev)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cases = []
    # (name, fn, flops)
    def conv_case(name, H, W, ci, co, k, T_=1, stride=1, flags=0):
        x = rnd(N, H, W, ci)
        Ho, Wo = H // stride, W // stride
        out = torch.empty((N, Ho, Wo, co), device=dev)
        kd = 3 if T_ > 1 else 1
        w = rnd(co, ci, kd, k, k) * 0.05 if kd > 1 else rnd(co, ci, k, k) * 0.05
        wp = ops.pack_conv(w)
        b = rnd(co)
        pad = (1 if kd > 1 else 0, 1 if k > 1 else 0, 1 if k > 1 else 0)
        fl = 2.0 * N * Ho * Wo * co * ci * kd * k * k
        cases.append((name, lambda: ops.conv_igemm(x, wp, b, out, T=T_, k=(kd, k, k), stride=stride, pad=pad, cin=ci, cout=co,
                                                    flags=flags), fl))

    def wgrad_case(name, H, W, ci, co, k, T_=1):
        x = rnd(N, H, W, ci)
        g = rnd(N, H, W, co)
        kd = 3 if T_ > 1 else 1
        dw = torch.empty((co, ci, kd, k, k), device=dev)
        db = torch.empty(co, device=dev)
        pad = (1 if kd > 1 else 0, 1 if k > 1 else 0, 1 if k > 1 else 0)
        fl = 2.0 * N * H * W * co * ci * kd * k * k
        cases.append((name, lambda: ops.conv_wgrad(g, x, dw, db, T=T_, k=(kd, k, k), pad=pad, a_real=co, b_real=ci), fl))

    conv_case("conv3d_b fwd 128->128 @64^2", 64, 64, 128, 128, 3, T_=T)
    conv_case("conv3d_t fwd 128->128 @32^2", 32, 32, 128, 128, 3, T_=T)
    conv_case("conv2d k3 128->128 @64^2", 64, 64, 128, 128, 3)
    conv_case("conv2d k3 128->32 @64^2 (relu in/out)", 64, 64, 128, 32, 3, flags=ops.FO_IN_RELU | ops.FO_OUT_RELU)
    conv_case("conv2d k1 32->128 @64^2", 64, 64, 32, 128, 1)
    conv_case("conv2d k4s2 64->128 @128^2", 128, 128, 64, 128, 4, stride=2)
    wgrad_case("wgrad conv3d_b 128x128 @64^2", 64, 64, 128, 128, 3, T_=T)
    wgrad_case("wgrad conv2d k3 128x128 @64^2", 64, 64, 128, 128, 3)
    wgrad_case("wgrad k3 32x128 @64^2", 64, 64, 128, 32, 3)
    wgrad_case("wgrad k1 128x32 @64^2", 64, 64, 32, 128, 1)
    # VQ
    xq = rnd(N, 64, 64, 64) * 0.5
    emb = rnd(64, 512) * 0.5
    eT, en = ops.vq_prepare(emb)
    q = torch.empty_like(xq)
    stats = torch.zeros(1 + 512 + 512 * 64, device=dev)
    cases.append(("vq_assign+stats bottom (655360 vec)", lambda: ops.vq_assign(xq, eT, en, q, stats, True), 2.0 * N * 4096 * 64 * 512))
    for name, fn, fl in cases:
        if flt and flt not in name:
            continue
        ms = timeit(fn)
        print(f"{name:45s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s nominal")


if __name__ == "__main__":
    main()
