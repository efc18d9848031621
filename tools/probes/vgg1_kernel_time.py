"""fo_vgg_conv1_fused_bf16 alone against conv1_1 + conv1_2 (with the pool riding along) as two launches, 160 frames of 256x256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd import _lib, ops
bf = torch.bfloat16
N, H, W = 160, 256, 256
x8 = torch.zeros((N, H, W, 8), device="cuda", dtype=bf); x8[..., :3] = torch.randn((N, H, W, 3), device="cuda").to(bf)
w1p = torch.zeros((64, 8, 3, 3), device="cuda"); w1p[:, :3] = torch.randn((64, 3, 3, 3), device="cuda") * 0.3
wp1, wp2 = ops.pack_conv_bf16(w1p, taps_pad=16), ops.pack_conv_bf16(torch.randn((64, 64, 3, 3), device="cuda") * 0.06)
b1, b2 = torch.randn(64, device="cuda") * 0.1, torch.randn(64, device="cuda") * 0.1
o1 = torch.empty((N, H, W, 64), device="cuda", dtype=bf); o2 = torch.empty_like(o1); pl = torch.empty((N, H // 2, W // 2, 64), device="cuda", dtype=bf)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def fused(keep):
    _lib.call("fo_vgg_conv1_fused_bf16", ops._ptr(x8), ops._ptr(wp1), ops._ptr(b1), ops._ptr(wp2), ops._ptr(b2), ops._ptr(o1) if keep else None, ops._ptr(o2),
              ops._ptr(pl), N, H, W, ops._stream())

def two():
    ops.conv_bf16(x8, wp1, b1, o1, cin=8, cout=64, flags=ops.FO_OUT_RELU)
    ops.conv_bf16(o1, wp2, b2, o2, cin=64, cout=64, flags=ops.FO_OUT_RELU, pooled=pl)

def rgb():
    ops.conv_bf16(x8, wp1, b1, o1, cin=8, cout=64, flags=ops.FO_OUT_RELU)

for rep in range(2):
    print(f"fused (no relu1_1 out) {timeit(lambda: fused(False)):.3f} ms; fused (+ relu1_1 out) {timeit(lambda: fused(True)):.3f} ms; two launches {timeit(two):.3f} ms "
          f"(conv1_1 alone {timeit(rgb):.3f})", flush=True)
