"""Launch times of the uint8 perturbation kernels on a loader batch (160 frames 256x256x3).  python tools/probes/u8_perturb_bench.py"""
import random
import torch
from faceoff_amd import perturbations as P

N = 160
x = torch.randint(0, 256, (N, 256, 256, 3), dtype=torch.uint8, device="cuda")
r = random.Random(0)
rot = [r.randint(-25, 25) for _ in range(N)]
mag = [r.randint(90, 110) / 100 for _ in range(N)]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


mb = 2 * x.numel() / 1e6
for name, fn in (("rotate (per-frame)", lambda: P.rotate_image(rot, x)), ("resize (per-frame)", lambda: P.resize_image(mag, x)),
                 ("flip", lambda: P.image_flip(1, x)), ("to_normalized", lambda: P.to_normalized(x))):
    ms = timed(fn)
    print(f"{name:20s} {ms * 1e3:8.1f} us  ({mb / ms / 1e3:.2f} TB/s of u8 in + u8 out)" if name != "to_normalized" else
          f"{name:20s} {ms * 1e3:8.1f} us  ({(x.numel() * 5) / 1e6 / ms / 1e3:.2f} TB/s of u8 in + f32 out)")
