"""Race screen for the split-bf16 GEMM kernels: the same launch repeated, alone and beside an HBM-heavy kernel on another stream,
must give the same bits every time (single-barrier K-steps with LDS-DMA + LDS writes in flight: a missing wait shows up here).
python tools/race_screen_split.py [reps]      (FACEOFF_SPLIT_NO256=1 for the 128-row kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceoff_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SHAPES = [(36, 160, 5, 256, 128, 128, 3), (36, 160, 5, 64, 128, 128, 3), (36, 160, 1, 256, 128, 128, 1), (25, 1, 1, 160 * 64, 256, 128, 1),
          (36, 10, 5, 64, 64, 256, 3), (25, 1, 1, 1152, 512, 128, 1)]
bad = 0
for planes, N, T, P, cin, cout, kd in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(planes + cin)
    V = torch.randn((planes, N * P, cin), device="cuda", generator=g)
    U = torch.randn((planes, cout, kd * cin), device="cuda", generator=g) * 0.05
    ws = torch.empty(3 * planes * cout * kd * cin, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    ref = None
    for r in range(reps):
        M = torch.full((planes, N * P, cout), float("nan"), device="cuda")
        if r % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        _lib.call("fo_wino_gemm_split", V.data_ptr(), U.data_ptr(), ws.data_ptr(), M.data_ptr(), planes, N, T, P, cin, cout, kd,
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = M
            assert torch.isfinite(ref).all()
        elif not torch.equal(M, ref):
            bad += 1
            print("MISMATCH", (planes, N, T, P, cin, cout, kd), "rep", r, int((M != ref).sum()))
    print((planes, N, T, P, cin, cout, kd), "ok" if bad == 0 else "BAD")
# ---- the filter-gradient kernels (per-tap and one-pass): dU must be bit-identical launch to launch (slabs summed in a fixed order)
import ctypes as C
for planes, N, T, P, cin, cout, kd in [(36, 160, 5, 256, 128, 128, 3), (36, 160, 5, 64, 128, 128, 3), (36, 160, 1, 256, 128, 128, 1),
                                       (25, 1, 1, 160 * 64, 256, 128, 1), (5, 6, 3, 32, 128, 256, 3)]:
    g = torch.Generator(device="cuda").manual_seed(planes + cin + 1)
    dM = torch.randn((planes, N * P, cout), device="cuda", generator=g)
    V = torch.randn((planes, N * P, cin), device="cuda", generator=g)
    nb = _lib.load().fo_wino_wgrad_split_ws_bytes(planes, N, P, cin, cout, kd)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    ref = None
    for r in range(reps):
        ws = torch.full((nb // 4,), float("nan"), device="cuda")
        dU = torch.full((planes, cout, cin, kd), float("nan"), device="cuda")
        if r % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        _lib.call("fo_wino_wgrad_split", dM.data_ptr(), V.data_ptr(), dU.data_ptr(), ws.data_ptr(), C.c_int64(nb), planes, N, T, P, cin, cout, kd,
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = dU
            assert torch.isfinite(ref).all()
        elif not torch.equal(dU, ref):
            bad += 1
            print("MISMATCH wgrad", (planes, N, T, P, cin, cout, kd), "rep", r, int((dU != ref).sum()))
    print("wgrad", (planes, N, T, P, cin, cout, kd), "ok" if bad == 0 else "BAD")
print("mismatching launches:", bad)
sys.exit(1 if bad else 0)
