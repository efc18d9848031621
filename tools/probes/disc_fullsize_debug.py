"""Isolate the discriminators' input gradient at the benched size from the generator: random fake / real inputs, generator-form RaLSGAN loss,
d loss / d fake from DiscEngine against the oracle's autograd -- per scale."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd.synth import make_disc_state
from faceoff_amd.disc import DiscEngine, ralsgan_pair
from oracle import disc_oracle as D

torch.set_num_threads(16)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Fp = int(sys.argv[2]) if len(sys.argv) > 2 else 15


def cl(x):
    if x.dim() == 4:
        x = x.unsqueeze(2)
    N, Cc, Dd, H, W = x.shape
    out = torch.zeros((N, Dd, H, W, 32), device="cuda")
    out[..., :Cc] = x.permute(0, 2, 3, 4, 1).cuda()
    return out.contiguous()


for dims in (3, 2):
    rng = np.random.default_rng(3)
    shape = (1, 6, Fp, size, size) if dims == 3 else (1, 6, size, size)
    real = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake0 = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    sd = make_disc_state(1 if dims == 3 else 2, dims)
    def engine_masks(S, sample):
        out = []
        for sc in S["scales"]:
            per = []
            for j in range(4):
                f = sc["feat"][j][sample]
                co = (64, 128, 256, 512)[j]
                m = (f[..., :co] > 0).permute(3, 0, 1, 2).cpu()
                per.append((m if dims == 3 else m[:, 0]).unsqueeze(0))
            out.append(per)
        return out
    eng0 = DiscEngine(sd, "cuda:0", dims=dims, n_frames=Fp)
    S0 = eng0.forward(torch.cat([cl(fake0), cl(real)], 0), training=True, sample_order=[0, 1])
    mf, mr = engine_masks(S0, 0), engine_masks(S0, 1)
    for which in ("both", "both forced", "scale0 forced", "scale1 forced"):
        fake = fake0.clone().requires_grad_(True)
        p = D.to_torch_state(sd)
        forced = "forced" in which
        Df = D.multiscale_discriminator(fake, p, n_frames=Fp, buffers={}, force_masks=mf if forced else None)
        Dr = D.multiscale_discriminator(real, p, n_frames=Fp, buffers={}, force_masks=mr if forced else None)
        if forced:
            print("   mask differences fake:", [D.mask_differences(Df[i], mf[i]) for i in range(2)], "real:", [D.mask_differences(Dr[i], mr[i]) for i in range(2)])
        sel = (0, 1) if which.startswith("both") else ((0,) if which.startswith("scale0") else (1,))
        loss = sum(D.ralsgan([Df[i]], [Dr[i]], True) + D.ralsgan([Dr[i]], [Df[i]], False) for i in sel) * 0.5
        loss.backward()
        eng = DiscEngine(sd, "cuda:0", dims=dims, n_frames=Fp)
        x = torch.cat([cl(fake0), cl(real)], 0)
        S = eng.forward(x, training=True, sample_order=[0, 1])
        l = torch.zeros(1, device="cuda")
        g = ralsgan_pair(S["logits"], 0, 1, 1.0, 0.0, 0.5, l, want_gb=False)
        for i in range(2):
            if i not in sel:
                g[i].zero_()
        gx = eng.backward(S, g, param_grads=False, input_grad=True, samples=(0, 1))
        torch.cuda.synchronize()
        got = gx[0, ..., :6].permute(3, 0, 1, 2).reshape(fake0.shape).cpu()
        want = fake.grad
        err = (got - want).abs()
        print(f"dims={dims} {which}: oracle loss {loss.item():.7f}; d/dfake max err {err.max().item() / want.abs().max().item():.3e} of max, rel-L2 {((got - want).norm() / want.norm()).item():.3e}; "
              f"where: frame/row/col of the max err {np.unravel_index(int(err.argmax()), err.shape)}")
