import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd.synth import make_disc_state
from faceoff_amd.disc import DiscEngine, ralsgan_pair
from oracle import disc_oracle as D
torch.set_num_threads(16)

def cl(x):
    x = x.unsqueeze(2)
    N, Cc, Dd, H, W = x.shape
    out = torch.zeros((N, Dd, H, W, 32), device="cuda")
    out[..., :Cc] = x.permute(0, 2, 3, 4, 1).cuda()
    return out.contiguous()

sd = make_disc_state(2, 2)
for size in (64, 128, 192, 256):
    rng = np.random.default_rng(3)
    shape = (1, 6, size, size)
    real = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake0 = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
    fake = fake0.clone().requires_grad_(True)
    p = D.to_torch_state(sd)
    Df = D.multiscale_discriminator(fake, p, buffers={})
    Dr = D.multiscale_discriminator(real, p, buffers={})
    for f in Df[0]:
        f.retain_grad()
    loss = (D.ralsgan([Df[0]], [Dr[0]], True) + D.ralsgan([Dr[0]], [Df[0]], False)) * 0.5
    loss.backward()
    eng = DiscEngine(sd, "cuda:0", dims=2)
    x = torch.cat([cl(fake0), cl(real)], 0)
    S = eng.forward(x, training=True, sample_order=[0, 1])
    l = torch.zeros(1, device="cuda")
    g = ralsgan_pair(S["logits"], 0, 1, 1.0, 0.0, 0.5, l, want_gb=False)
    g[1].zero_()
    gx = eng.backward(S, g, param_grads=False, input_grad=True, samples=(0, 1))
    torch.cuda.synchronize()
    got = gx[0, ..., :6].permute(3, 0, 1, 2).reshape(fake0.shape).cpu()
    want = fake.grad
    err = (got - want).abs() / want.abs().max()
    e2 = err[0].amax(0)
    bad = e2 > 1e-3
    ys, xs = np.nonzero(bad.numpy())
    # forward features vs oracle
    ff = []
    for j in range(5):
        f = S["scales"][0]["feat"][j][0, 0]                   # [H,W,C] of sample 0
        w_ = Df[0][j][0].detach().permute(1, 2, 0)
        c = w_.shape[-1]
        ff.append(float((f[..., :c].cpu() - w_).abs().max() / w_.abs().max()))
    for j in range(4):
        f = S["scales"][0]["feat"][j][0, 0]
        w_ = Df[0][j][0].detach().permute(1, 2, 0)
        c = w_.shape[-1]
        fl = ((f[..., :c].cpu() > 0) != (w_ > 0))
        if fl.any():
            idx = fl.nonzero()
            print(f"   layer {j}: {int(fl.sum())} LeakyReLU sign differences at (y, x, c) {idx[:3].tolist()}; |value| there: engine {f[..., :c].cpu()[fl].abs().max().item():.3e}, oracle {w_[fl].abs().max().item():.3e} "
                  f"(layer scale {w_.abs().max().item():.2f})")
    print(f"size {size}: loss {l.item():.6f} vs {loss.item():.6f}; fwd feature max errs {['%.1e' % v for v in ff]}; d/dfake rel-L2 {((got - want).norm() / want.norm()).item():.3e}, max {err.max().item():.3e}; "
          f"pixels > 1e-3: {int(bad.sum())} of {bad.numel()}; rows {ys.min() if len(ys) else -1}..{ys.max() if len(ys) else -1} cols {xs.min() if len(xs) else -1}..{xs.max() if len(xs) else -1}")
