#!/usr/bin/env python3
"""The k4 s2 p1 stems at the C2 sizes (160 frames): direct kernels vs the Winograd F(4x4, 2x2) form, per pass, interleaved in one
process (HIP events, median of 7 rounds x 3 launches):  python tools/bench_w42.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("FRAMES", "160"))


def ab(fns, rounds=7, reps=3):
    ts = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, fn in fns.items():
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            if r:
                ts[k].append(s.elapsed_time(e) / reps)
    return {k: statistics.median(v) for k, v in ts.items()}


def conv_layer(name, cin, cout, H):
    """Conv2d k4 s2 p1 cin -> cout on H x H frames: forward, data gradient (transposed form), filter gradient"""
    x = torch.randn((N, H, H, cin), device=dev)
    w = torch.randn((cout, cin, 4, 4), device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty((N, H // 2, H // 2, cout), device=dev)
    g = torch.randn_like(y)
    gx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
    wp, wpd = ops.pack_conv(w), ops.pack_convT(w)
    U, Ut = ops.w42_filter(w, False), ops.w42_filter(w, True)
    flop = 2.0 * N * (H // 2) ** 2 * cout * cin * 16
    fwd_ok = ops.w42_conv_ok(N, H, H, cin, cout)
    if fwd_ok:
        r = ab({"direct": lambda: ops.conv_igemm(x, wp, b, y, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=cin, cout=cout, flags=ops.FO_OUT_RELU),
                "w42": lambda: ops.conv_k4s2_winograd(x, U, b, y, cin=cin, cout=cout, flags=ops.FO_OUT_RELU)})
        print(f"{name} fwd    direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42 {r['w42']:.3f} ms", flush=True)
    r = ab({"direct": lambda: ops.convT_phases(g, wpd, None, gx, cin=cout, cout=cin),
            "w42": lambda: ops.convT_k4s2_winograd(g, Ut, None, gx, cin=cout, cout=cin)})
    print(f"{name} dgrad  direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42 {r['w42']:.3f} ms", flush=True)
    V = ops.conv_k4s2_winograd(x, U, b, y, cin=cin, cout=cout, keep_v=True) if fwd_ok else None
    r = ab({"direct": lambda: ops.conv_wgrad(g, x, dw, db, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=cout, b_real=cin),
            "w42 (V kept)": lambda: ops.conv_k4s2_wgrad_winograd(x, g, dw, cin=cin, cout=cout, V=V),
            "w42 (V recomputed)": lambda: ops.conv_k4s2_wgrad_winograd(x, g, dw, cin=cin, cout=cout)})
    print(f"{name} wgrad  direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42, V kept {r['w42 (V kept)']:.3f} ms   "
          f"w42, V recomputed {r['w42 (V recomputed)']:.3f} ms", flush=True)


def convT_layer(name, cin, cout, h):
    """ConvTranspose2d k4 s2 p1 cin -> cout on h x h frames: forward (transposed form), data gradient (conv form), filter gradient"""
    x = torch.randn((N, h, h, cin), device=dev)
    w = torch.randn((cin, cout, 4, 4), device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty((N, 2 * h, 2 * h, cout), device=dev)
    g = torch.randn_like(y)
    gx = torch.empty_like(x)
    dw = torch.empty_like(w)
    wp, wpd = ops.pack_convT(w), ops.pack_conv(w)
    Ut, U = ops.w42_filter(w, True), ops.w42_filter(w, False)
    flop = 2.0 * N * h * h * cout * cin * 16
    r = ab({"direct": lambda: ops.convT_phases(x, wp, b, y, cin=cin, cout=cout, flags=ops.FO_OUT_RELU),
            "w42": lambda: ops.convT_k4s2_winograd(x, Ut, b, y, cin=cin, cout=cout, flags=ops.FO_OUT_RELU)})
    print(f"{name} fwd    direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42 {r['w42']:.3f} ms", flush=True)
    if ops.w42_conv_ok(N, 2 * h, 2 * h, cout, cin):
        r = ab({"direct": lambda: ops.conv_igemm(g, wpd, None, gx, k=(1, 4, 4), stride=2, pad=(0, 1, 1), cin=cout, cout=cin),
                "w42": lambda: ops.conv_k4s2_winograd(g, U, None, gx, cin=cout, cout=cin)})
        print(f"{name} dgrad  direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42 {r['w42']:.3f} ms", flush=True)
    r = ab({"direct": lambda: ops.conv_wgrad(x, g, dw, None, k=(1, 4, 4), stride=2, pad=(0, 1, 1), a_real=cin, b_real=cout),
            "w42": lambda: ops.conv_k4s2_wgrad_winograd(g, x, dw, cin=cout, cout=cin)})
    print(f"{name} wgrad  direct {r['direct']:.3f} ms ({flop / r['direct'] / 1e9:.0f} TF)   w42 {r['w42']:.3f} ms", flush=True)


conv_layer("enc_b.2  64->128 @128", 64, 128, 128)
conv_layer("enc_t.0 128->64  @64 ", 128, 64, 64)
convT_layer("dec.4   128->64  @64 ", 128, 64, 64)
convT_layer("dec_t.4 128->64  @32 ", 128, 64, 32)
convT_layer("upsmp_t  64->64  @32 ", 64, 64, 32)
