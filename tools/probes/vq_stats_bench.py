"""vq_stats launch time at the C2 sizes, uniform and skewed assignments.  PYTHONPATH=. python tools/probes/vq_stats_bench.py"""
import torch
from faceoff_amd import ops, _lib
import ctypes as C
dev = torch.device("cuda:0")
for nvec, name in ((160 * 64 * 64, "bottom 64x64"), (160 * 32 * 32, "top 32x32")):
    x = torch.randn(nvec, 64, device=dev)
    for dist in ("uniform", "zipf", "one code"):
        if dist == "uniform":
            ind = torch.randint(0, 512, (nvec,), device=dev)
        elif dist == "zipf":
            p = 1.0 / torch.arange(1, 513, dtype=torch.float64)
            ind = torch.multinomial((p / p.sum()).float(), nvec, replacement=True).to(dev)
        else:
            ind = torch.full((nvec,), 37, device=dev, dtype=torch.int64)
        counts, esum = torch.empty(512, device=dev), torch.empty(512 * 64, device=dev)
        nb = _lib.load().fo_vq_stats_ws_bytes(C.c_int64(nvec))
        ws = torch.empty(nb // 4 + 16, device=dev)

        def run():
            _lib.call("fo_vq_stats", ops._ptr(x), 64, C.c_int64(nvec), ops._ptr(ind), ops._ptr(counts), ops._ptr(esum), ops._ptr(ws), ops._stream())
        try:
            run()
        except Exception as e:
            print("call failed:", e); raise
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record(); torch.cuda.synchronize()
        print(f"{name:14s} {dist:9s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us")
