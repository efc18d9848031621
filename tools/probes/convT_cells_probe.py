"""What ONE launch per k4 s2 transposed layer could give (bf16 engine): the four sub-pixel phase launches of the layer as they run today against a
single GEMM-shaped launch with the cell form's geometry (k2 full correlation over the (H+1) x (W+1) cell grid, K = 4 Cin, N = 4 Cout columns; plain
epilogue, values meaningless: timing only).     python tools/probes/convT_cells_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd import ops  # noqa: E402

bf = torch.bfloat16
N = 160


def timeit(fn, reps=10):
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


for name, H, cin, cout in (("dec.blocks.4 / enc_b.2 dgrad", 64, 128, 64), ("dec_t.blocks.4", 32, 128, 64), ("upsample_t", 32, 64, 64), ("enc_t.0 dgrad", 32, 64, 128)):
    x = (torch.randn((N, H, H, cin), device="cuda") * 0.5).to(bf)
    w = torch.randn((cin, cout, 4, 4), device="cuda") * 0.05
    wp4 = ops.to_bf16(ops.pack_convT(w))
    out = torch.empty((N, 2 * H, 2 * H, cout), device="cuda", dtype=bf)
    b = torch.randn(cout, device="cuda")
    t4 = timeit(lambda: ops.convT_phases_bf16(x, wp4, b, out, cin=cin, cout=cout, flags=ops.FO_OUT_RELU))
    # the cell-form GEMM: k2, pad 1 on the (H+1) x (W+1) grid, 4 * cout columns
    wc = ops.pack_conv_bf16(torch.randn((4 * cout, cin, 2, 2), device="cuda") * 0.05)
    outc = torch.empty((N, H + 1, H + 1, 4 * cout), device="cuda", dtype=bf)
    bc = torch.randn(4 * cout, device="cuda")
    t1 = timeit(lambda: ops.conv_bf16g(x, wc, bc, outc, k=(1, 2, 2), stride=1, pad=(0, 1, 1), cin=cin, cout=4 * cout, flags=ops.FO_OUT_RELU, mgrid=(H + 1, H + 1)))
    gf = 2.0 * N * H * H * 16 * cin * cout / 1e9
    print(f"{name:32s} {N}x{H}x{H} {cin}->{cout}: 4 phase launches {t4:.3f} ms ({gf / t4:.0f} TFLOP/s); one cell-form GEMM launch {t1:.3f} ms ({gf / t1:.0f} TFLOP/s)")
