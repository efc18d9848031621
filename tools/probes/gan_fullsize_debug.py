"""Where does the generator iteration's gradient error at the benched config-5 size come from?  (1) the VQ-VAE step alone on the 30-frame clip vs the
oracle; (2) the generator iteration; (3) the same with the engine on the direct kernels.  Top gradient errors each.   gpurun -- python tools/probes/gan_fullsize_debug.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faceoff_amd.synth import make_state_dict, make_batch, make_disc_state
from faceoff_amd.disc import DiscEngine
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd.gan_trainer import GANTrainer
from faceoff_amd import ops
from oracle import disc_oracle as D
from oracle import faceoff_oracle as O

torch.set_num_threads(16)
n, h, w, win = 30, 256, 256, 16
sd, sd3, sd2 = make_state_dict(0, codebook_scale=0.3, gain=2.0), make_disc_state(1, 3), make_disc_state(2, 2)
img, gt = make_batch(55, 1, n, h, w)
x_img = torch.from_numpy(img).reshape(n, 6, h, w).cuda()
x_gt = torch.from_numpy(gt).reshape(n, 3, h, w).cuda()
c = dict(random_idx=5, frame_id=7, flip_real=True, flip_fake=False)


def errs(got, want):
    out = []
    for k, wv in want.items():
        scale = wv.abs().max().item() + 1e-30
        out.append((float((got[k].cpu() - wv).abs().max().item() / scale), k))
    return sorted(out, reverse=True)


def oracle(gan):
    p = O.to_torch_state(sd)
    fw = O.vqvae_forward(torch.from_numpy(img), p, training=True)
    fw["dec"].retain_grad()
    out = fw["dec"][:, :3]
    gtt = torch.from_numpy(gt).reshape(n, 3, h, w)
    loss = torch.nn.functional.mse_loss(out, gtt) + fw["diff"].mean()
    if gan:
        r = c["random_idx"]
        g2d, g3d = D.generator_gan_losses(out[r:r + win].unsqueeze(0), gtt[r:r + win].unsqueeze(0), D.to_torch_state(sd3), D.to_torch_state(sd2),
                                          c["frame_id"], c["flip_real"], c["flip_fake"], {}, {})
        loss = loss + g2d + g3d
    loss.backward()
    return p, fw


def engine_masks(S, sample, dims):
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

with torch.no_grad():
    fw0 = O.vqvae_forward(torch.from_numpy(img), O.to_torch_state(sd), training=True)
ids = (fw0["id_t"].cuda(), fw0["id_b"].cuda())
eng = VQVAEEngine(sd, "cuda:0")
tr = GANTrainer(eng, DiscEngine(sd3, "cuda:0", dims=3, n_frames=win - 1), DiscEngine(sd2, "cuda:0", dims=2), lr=3e-4, d_lr=1e-4, window=win)
tr.optimizer.step = lambda grad_scale=1.0: None
g_dec_keep = {}
orig_bwd = eng.backward
def bwd(S, g_dec, lw):
    g_dec_keep["g"] = g_dec.clone()
    return orig_bwd(S, g_dec, lw)
eng.backward = bwd
o = tr.step(x_img, x_gt, c, force_ids=ids)
torch.cuda.synchronize()
S2, S3 = tr.last_disc_states
masks = dict(fake2=engine_masks(S2, 0, 2), real2=engine_masks(S2, 1, 2), fake3=engine_masks(S3, 0, 3), real3=engine_masks(S3, 1, 3))
# the GAN part of d loss / d dec, from the fp32 oracle with the engine's LeakyReLU branches (it equals the engine's to 5e-6)
p = O.to_torch_state(sd)
fw = O.vqvae_forward(torch.from_numpy(img), p, training=True, force_ids=(fw0["id_t"], fw0["id_b"]))
fw["dec"].retain_grad()
out = fw["dec"][:, :3]
gtt = torch.from_numpy(gt).reshape(n, 3, h, w)
r = c["random_idx"]
g2d, g3d = D.generator_gan_losses(out[r:r + win].unsqueeze(0), gtt[r:r + win].unsqueeze(0), D.to_torch_state(sd3), D.to_torch_state(sd2), c["frame_id"], c["flip_real"], c["flip_fake"], {}, {}, masks=masks)
(g2d + g3d).backward(retain_graph=True)
g_gan = fw["dec"].grad.detach().clone()
for v in p.values():
    v.grad = None
# (a) the fp32 oracle's parameter gradients for  mse + latent + <dec, g_gan>
loss32 = torch.nn.functional.mse_loss(out, gtt) + fw["diff"].mean() + (fw["dec"] * g_gan).sum()
loss32.backward()
g32 = {k: v.grad.clone() for k, v in p.items() if v.requires_grad}
# (b) the same function in fp64 (same codes)
p64 = {k: (v.detach().double().requires_grad_(v.requires_grad)) for k, v in O.to_torch_state(sd).items()}
fw64 = O.vqvae_forward(torch.from_numpy(img).double(), p64, training=True, force_ids=(fw0["id_t"], fw0["id_b"]))
loss64 = torch.nn.functional.mse_loss(fw64["dec"][:, :3], gtt.double()) + fw64["diff"].mean() + (fw64["dec"] * g_gan.double()).sum()
loss64.backward()
g64 = {k: v.grad.float() for k, v in p64.items() if v.requires_grad}
e_eng64, e_3264, e_eng32 = errs(eng.grads, g64), errs({k: v for k, v in g32.items()}, g64), errs(eng.grads, g32)
class _G(dict):
    pass
print("engine vs fp64 oracle: top", [(round(a_, 6), b_) for a_, b_ in e_eng64[:3]], "median %.2e" % e_eng64[len(e_eng64) // 2][0])
print("fp32 oracle vs fp64 oracle: top", [(round(a_, 6), b_) for a_, b_ in errs({k: v.cuda() for k, v in g32.items()}, g64)[:3]], "median %.2e" % errs({k: v.cuda() for k, v in g32.items()}, g64)[35][0])
print("engine vs fp32 oracle: top", [(round(a_, 6), b_) for a_, b_ in e_eng32[:3]], "median %.2e" % e_eng32[len(e_eng32) // 2][0])
