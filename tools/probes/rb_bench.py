import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faceoff_amd import ops
N, H = 160, 64
x = torch.randn((N, H, H, 128), device="cuda")
w1, b1 = torch.randn((32, 128, 3, 3), device="cuda") * 0.05, torch.randn(32, device="cuda")
w3, b3 = torch.randn((128, 32, 1, 1), device="cuda") * 0.1, torch.randn(128, device="cuda")
wp1, wp3 = ops.pack_conv(w1), ops.pack_conv(w3)
hb = torch.empty((N, H, H, 32), device="cuda"); out = torch.empty((N, H, H, 128), device="cuda")
for _ in range(3): ops.resblock_fwd(x, wp1, b1, wp3, b3, hb, out, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ops.resblock_fwd(x, wp1, b1, wp3, b3, hb, out, False)
torch.cuda.synchronize(); print(f"resblock_fwd 64^2: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
