#!/usr/bin/env python3
"""Per-layer timing of the video discriminator's kernels at the C5 sizes (2 samples x 15 frame pairs x 256x256):
    python tools/bench_disc.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import _lib, ops
from faceoff_amd._lib import ConvNdDesc
dev = torch.device("cuda:0")

def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

N = 2
chans = [(32, 6, 64, 2), (64, 64, 128, 2), (128, 128, 256, 2), (256, 256, 512, 1), (512, 512, 1, 1)]
for scale, (D, H, W) in enumerate([(15, 256, 256), (15, 128, 128)]):
    tot = 0.0
    for j, (cs, cin_real, co, s) in enumerate(chans):
        Do, Ho, Wo = (D + 4 - 4) // s + 1, (H + 4 - 4) // s + 1, (W + 4 - 4) // s + 1
        ldo = max(32, co)
        x = torch.randn((N, D, H, W, cs), device=dev)
        w = torch.randn((co, cin_real, 64), device=dev) * 0.02
        wp = torch.empty(((co + 63) // 64 * 64) * 64 * cs, device=dev)
        _lib.call("fo_pack_convnd", ops._ptr(w), ops._ptr(wp), co, cin_real, 64, 0, ops._stream())
        wpt = torch.empty(((cin_real + 63) // 64 * 64) * 64 * ldo, device=dev)
        _lib.call("fo_pack_convnd", ops._ptr(w), ops._ptr(wpt), co, cin_real, 64, 1, ops._stream())
        y = torch.zeros((N, Do, Ho, Wo, ldo), device=dev)
        g = torch.randn_like(y)
        gin = torch.empty_like(x)
        dw = torch.zeros_like(w)
        d = ConvNdDesc(N=N, Ds=D, Hs=H, Ws=W, Cs=cs, ldS=cs, Dd=Do, Hd=Ho, Wd=Wo, Cd=co, ldD=ldo, KD=4, KH=4, KW=4, sD=s, sH=s, sW=s, pD=2, pH=2, pW=2,
                       ldMask=0, flags=0, slope=0.2)
        dt = ConvNdDesc(N=N, Ds=Do, Hs=Ho, Ws=Wo, Cs=ldo, ldS=ldo, Dd=D, Hd=H, Wd=W, Cd=cin_real, ldD=cs, KD=4, KH=4, KW=4, sD=s, sH=s, sW=s, pD=2, pH=2,
                        pW=2, ldMask=0, flags=0, slope=0.2)
        flop = 2.0 * N * Do * Ho * Wo * co * cin_real * 64
        t_f = timeit(lambda: _lib.call("fo_convnd", C.byref(d), 0, ops._ptr(x), ops._ptr(wp), None, None, ops._ptr(y), None, C.c_int64(0), ops._stream()))
        t_d = timeit(lambda: _lib.call("fo_convnd", C.byref(dt), 1, ops._ptr(g), ops._ptr(wpt), None, None, ops._ptr(gin), None, C.c_int64(0), ops._stream()))
        ws = ops._workspace(_lib.load().fo_wgradnd_ws_bytes(C.byref(d)), dw.device)
        t_w = timeit(lambda: _lib.call("fo_wgradnd", C.byref(d), ops._ptr(g), ops._ptr(x), ops._ptr(dw), cin_real, ops._ptr(ws), C.c_int64(ws.numel() * 4),
                                       ops._stream()))
        tot += t_f + t_d + t_w
        print(f"scale {scale} layer {j}: {cin_real:3d}->{co:3d} s{s} out {Do}x{Ho}x{Wo}  {flop / 1e9:7.1f} GFLOP   fwd {t_f:7.3f} ms ({flop / t_f / 1e9:6.1f} TF)   "
              f"dgrad {t_d:7.3f} ms ({flop / t_d / 1e9:6.1f} TF)   wgrad {t_w:7.3f} ms ({flop / t_w / 1e9:6.1f} TF)")
        D, H, W = Do, Ho, Wo
    print(f"scale {scale} total {tot:.2f} ms")
