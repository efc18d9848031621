#!/usr/bin/env python3
"""Per-step wall time and allocator state of the C3 trainer over many steps: python tools/probes/c3_steps_probe.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.trainer import FaceOffTrainer
from faceoff_amd.loss import VQLPIPS
from faceoff_amd.synth import make_state_dict, make_vgg_lpips_state
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gen = torch.Generator(device=dev).manual_seed(1234)
src = torch.rand((160, 3, 256, 256), device=dev, generator=gen) * 2 - 1
bg = torch.rand((160, 3, 256, 256), device=dev, generator=gen) * 2 - 1
gt = torch.rand((160, 3, 256, 256), device=dev, generator=gen) * 2 - 1
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev, dtype="bf16")
tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(make_vgg_lpips_state(7), dtype="bf16").to(dev))
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step((src, bg), gt, T=5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    print(f"step {i:2d}: {dt:7.2f} ms  allocated {torch.cuda.memory_allocated()/2**30:6.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:6.2f} GiB "
          f"mallocs {st.get('num_device_alloc', 0)} frees {st.get('num_device_free', 0)} retries {st.get('num_alloc_retries', 0)}")
# the same, the host running ahead (no synchronisation between steps): does the pool grow?
st0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    tr.step((src, bg), gt, T=5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / steps
st = torch.cuda.memory_stats()
print(f"async x{steps}: {dt:7.2f} ms/step  reserved {torch.cuda.memory_reserved()/2**30:6.2f} GiB  new device mallocs {st.get('num_device_alloc', 0) - st0} retries {st.get('num_alloc_retries', 0)}")
