#!/usr/bin/env python3
"""Per-launch time and rate of the discriminators' generic convolutions (fo_convnd) inside a GAN iteration: python tools/probes/gan_conv_probe.py"""
import os, sys, random, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd import _lib
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
tr.gen.set_stream_overlap(False) if hasattr(tr, "gen") else None
for _ in range(3):
    tr.step(img, gt)
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = _lib.call
def timed(name, *args):
    if name not in ("fo_convnd", "fo_wgradnd"):
        return orig(name, *args)
    d = args[0]._obj
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); orig(name, *args); e1.record(); torch.cuda.synchronize()
    f = [getattr(d, n) for n, _ in d._fields_]
    key = (name, args[1] if name == "fo_convnd" else args[4], tuple(f[:-1]))
    rec.setdefault(key, []).append(e0.elapsed_time(e1))
_lib.call = timed
import faceoff_amd.disc as disc
disc._lib.call = timed
for _ in range(2):
    tr.step(img, gt)
_lib.call = orig
names = [n for n, _ in _lib.ConvNdDesc._fields_]
print(names)
tot = 0
for (name, mode, f), ts in sorted(rec.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(ts) / 2
    tot += ms
    d = dict(zip(names, f))
    if name == "fo_convnd" and mode == 0:     # forward: src -> dst
        flop = 2.0 * d["N"] * d["Dd"] * d["Hd"] * d["Wd"] * d["Cd"] * d["KD"] * d["KH"] * d["KW"] * d["Cs"]
    elif name == "fo_convnd":                 # data gradient: dst-shaped gradient -> src (mode 1 swaps roles: Ds.. is the gradient's grid here)
        flop = 2.0 * d["N"] * d["Ds"] * d["Hs"] * d["Ws"] * d["Cs"] * d["KD"] * d["KH"] * d["KW"] * d["Cd"] / (d["sD"] * d["sH"] * d["sW"])
    else:
        flop = 2.0 * d["N"] * d["Dd"] * d["Hd"] * d["Wd"] * d["Cd"] * d["KD"] * d["KH"] * d["KW"] * d["Cs"]
    avg = sum(ts) / len(ts)
    print(f"{ms:7.3f} ms/pair x{len(ts)/2:4.1f} {avg:7.3f} ms {flop/avg/1e9:7.1f} TF  {name[3:]} m{mode} N{d['N']} src {d['Ds']}x{d['Hs']}x{d['Ws']}x{d['Cs']} dst {d['Dd']}x{d['Hd']}x{d['Wd']}x{d['Cd']} k{d['KD']}{d['KH']}{d['KW']} s{d['sD']}{d['sH']}{d['sW']} f{d['flags']}")
print("total", tot)
