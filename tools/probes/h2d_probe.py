import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.synth import make_state_dict
from faceoff_amd.trainer import FaceOffTrainer
dev = torch.device("cuda:0")
B, T, H = 32, 5, 256
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
tr = FaceOffTrainer(eng)
g = torch.Generator().manual_seed(1)
host = [tuple((torch.rand((B, T, 3, H, H), generator=g) * 2 - 1).pin_memory() if i != 4 else torch.empty(0) for i in range(5)) for _ in range(2)]
K = 6
# raw copy rate
d = torch.empty_like(host[0][0], device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): d.copy_(host[0][0], non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"H2D of {host[0][0].numel()*4/1e6:.0f} MB: {dt*1e3:.2f} ms = {host[0][0].numel()*4/dt/1e9:.1f} GB/s")
it = tr.run_host_fed([host[i % 2] for i in range(K + 2)])
next(it); next(it); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): next(it)
torch.cuda.synchronize(); print("fed ms/step", (time.perf_counter() - t0) / K * 1e3)
src, bg, gt = (host[0][i].reshape(-1, 3, H, H).to(dev) for i in (0, 2, 3))
for _ in range(2): tr.step((src, bg), gt, T=T)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): tr.step((src, bg), gt, T=T)
torch.cuda.synchronize(); print("resident ms/step", (time.perf_counter() - t0) / K * 1e3)
