"""Time fo_wino_gemm vs fo_wino_gemm_split at the C2 plane-stack shapes.  python tools/bench_wsplit.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceoff_amd import _lib

SHAPES = [  # planes, N, T, P, cin, cout, kd
    ("conv3d_b 128->128 @64^2", 36, 160, 5, 256, 128, 128, 3),
    ("conv3d_t 128->128 @32^2", 36, 160, 5, 64, 128, 128, 3),
    ("conv2d 3x3 128->128 @64^2", 36, 160, 1, 256, 128, 128, 1),
    ("w42 enc_b.2 256->128", 25, 1, 1, 160 * 64, 256, 128, 1),
]
for name, planes, N, T, P, cin, cout, kd in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    V = torch.randn((planes, N * P, cin), device="cuda")
    U = torch.randn((planes, cout, kd * cin), device="cuda") * 0.05
    M = torch.empty((planes, N * P, cout), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    ws = torch.empty(3 * planes * cout * kd * cin, device="cuda", dtype=torch.bfloat16)

    def call(sym):
        if sym == "fo_wino_gemm_split":
            _lib.call(sym, V.data_ptr(), U.data_ptr(), ws.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd, s)
        else:
            _lib.call(sym, V.data_ptr(), U.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd, s)

    for sym in ("fo_wino_gemm", "fo_wino_gemm_split"):
        for _ in range(3):
            call(sym)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call(sym)
        e1.record(); torch.cuda.synchronize()
        res[sym] = e0.elapsed_time(e1) / 10
    flop = 2.0 * planes * N * P * cin * kd * cout
    byts = 4.0 * planes * N * P * (cin + cout)
    print(f"{name:28s} fp32 {res['fo_wino_gemm']:.3f} ms ({flop / res['fo_wino_gemm'] / 1e9:.0f} TF)   bf16x6 {res['fo_wino_gemm_split']:.3f} ms"
          f" ({flop / res['fo_wino_gemm_split'] / 1e9:.0f} TF-equivalent, {byts / res['fo_wino_gemm_split'] / 1e9:.2f} TB/s algorithmic)")
