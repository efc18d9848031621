"""LPIPSEngine.features (the 13 VGG convolutions + pools on 160 frames of 256x256, bf16) with and without fo_vgg_conv1_fused_bf16:
    gpurun -- python tools/probes/vgg1_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd.lpips import LPIPSEngine
from faceoff_amd.synth import make_vgg_lpips_state
sd = make_vgg_lpips_state(7)
img = torch.rand((160, 3, 256, 256), device="cuda") * 2 - 1
for fuse in (True, False, True, False):
    eng = LPIPSEngine(sd, "cuda:0", dtype="bf16"); eng.fuse_conv1 = fuse
    for keep in (False, True):
        x8 = eng._prep(img, nhwc=False)
        eng.features(x8, keep_all=keep); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            eng.features(x8, keep_all=keep)
        e.record(); torch.cuda.synchronize()
        print("fuse", fuse, "keep_all", keep, "features ms", round(s.elapsed_time(e) / 3, 3), flush=True)
