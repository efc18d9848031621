#!/usr/bin/env python3
"""Per-launch-shape breakdown of the GAN iteration (BASELINE config 5) on one 30-frame clip, HIP events per profiled launch + a rocprofv3-free
estimate of what is NOT a profiled conv launch:  python tools/gan_breakdown.py"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd import ops
from faceoff_amd.disc import DiscEngine
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.gan_trainer import GANTrainer
from faceoff_amd.synth import make_state_dict, make_disc_state
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((30, 6, 256, 256), device=dev, generator=gen) * 2 - 1
gt = torch.rand((30, 3, 256, 256), device=dev, generator=gen) * 2 - 1
eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
tr = GANTrainer(eng, DiscEngine(make_disc_state(1, 3), dev, dims=3, n_frames=15), DiscEngine(make_disc_state(2, 2), dev, dims=2), rng=random.Random(3))
if "--overlap" not in sys.argv:
    eng.set_stream_overlap(False)
    tr.overlap_d2 = False
    tr.d3.overlap_scales = tr.d2.overlap_scales = False
for _ in range(4):
    tr.step(img, gt)
torch.cuda.synchronize()
for kind in ("generator", "discriminator"):
    if (tr.iteration % 2 == 0) != (kind == "generator"):
        tr.step(img, gt)
    prof = ops.KernelProfiler(detail=True)
    ops.PROFILER = prof
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(img, gt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    ops.PROFILER = None
    prof.close()                    # kernel notes off again (fo_kernel_notes(0)): they stay on for the rest of the process otherwise
    summ = prof.summary()
    tot = sum(v["total_ms"] for v in summ.values())
    print(f"== {kind} iteration {dt:.2f} ms; profiled conv launches {tot:.2f} ms; other {dt - tot:.2f} ms")
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:28]:
        print(f"{v['total_ms']:8.3f} ms  x{v['launches']:<3d} {v['avg_ms']:8.3f} ms  {v['tflops']:6.1f} TF  {k}")
