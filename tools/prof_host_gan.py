#!/usr/bin/env python3
"""cProfile of the host side of GAN iterations (config 5 is launch-bound):  python tools/prof_host_gan.py"""
import cProfile, os, pstats, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceoff_amd.disc import DiscEngine
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.gan_trainer import GANTrainer
from faceoff_amd.synth import make_state_dict, make_disc_state
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((30, 6, 256, 256), device=dev, generator=gen) * 2 - 1
gt = torch.rand((30, 3, 256, 256), device=dev, generator=gen) * 2 - 1
tr = GANTrainer(VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev), DiscEngine(make_disc_state(1, 3), dev, dims=3, n_frames=15),
                DiscEngine(make_disc_state(2, 2), dev, dims=2), rng=random.Random(3))
for _ in range(4):
    tr.step(img, gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    tr.step(img, gt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
