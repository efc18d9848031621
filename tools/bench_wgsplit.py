"""Time fo_conv_wgrad_banked vs fo_wino_wgrad_split at the C2 filter-gradient GEMM shapes.  python tools/bench_wgsplit.py [filter]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faceoff_amd import _lib
from faceoff_amd.ops import _desc

SHAPES = [  # planes, N, T, P, cin, cout, kd
    ("conv3d_b 128x128 @64^2", 36, 160, 5, 256, 128, 128, 3),
    ("conv3d_t 128x128 @32^2", 36, 160, 5, 64, 128, 128, 3),
    ("conv2d 3x3 128x128 @64^2", 36, 160, 1, 256, 128, 128, 1),
    ("w42 enc_b.2 128x256", 25, 1, 1, 160 * 64, 256, 128, 1),
]
for name, planes, N, T, P, cin, cout, kd in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    dM = torch.randn((planes, N * P, cout), device="cuda")
    V = torch.randn((planes, N * P, cin), device="cuda")
    dU = torch.empty((planes, cout, cin, kd), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    nb = _lib.load().fo_wino_wgrad_split_ws_bytes(planes, N, P, cin, cout, kd)
    ws = torch.empty(nb // 4, device="cuda")
    d = _desc(N=planes * N, T=T if kd > 1 else 1, Hin=1, Win=P, Hm=1, Wm=P, Hout=1, Wout=P, Cin=cin, Cout=cout, KD=kd, KH=1, KW=1, stride=1,
              padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=cin, ldOut=cout, ldMask=0, ldAdd=0, flags=0)
    nb2 = _lib.load().fo_wgrad_banked_ws_bytes(C.byref(d), planes)
    ws2 = torch.empty(nb2 // 4 + 16, device="cuda")

    def call(which):
        if which == "split":
            _lib.call("fo_wino_wgrad_split", dM.data_ptr(), V.data_ptr(), dU.data_ptr(), ws.data_ptr(), C.c_int64(nb), planes, N, T, P, cin, cout, kd, s)
        else:
            _lib.call("fo_conv_wgrad_banked", C.byref(d), dM.data_ptr(), V.data_ptr(), dU.data_ptr(), cout, cin, ws2.data_ptr(), C.c_int64(nb2), planes, s)
    res = {}
    for which in ("fp32", "split"):
        for _ in range(3):
            call(which)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call(which)
        e1.record(); torch.cuda.synchronize()
        res[which] = e0.elapsed_time(e1) / 10
    flop = 2.0 * planes * N * P * cin * kd * cout
    print(f"{name:28s} fp32 {res['fp32']:.3f} ms ({flop / res['fp32'] / 1e9:.0f} TF)   bf16x6 {res['split']:.3f} ms ({flop / res['split'] / 1e9:.0f} TF-equivalent)")
