"""Probe: the bf16-operand engine against the bf16-simulated oracle and against the fp32 oracle at a small size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from faceoff_amd.engine import VQVAEEngine
from faceoff_amd import ops
from faceoff_amd.synth import make_state_dict, make_batch
from oracle import faceoff_oracle as O

B, T, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 2, 64, 64)))
SEED = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sd = make_state_dict(SEED, codebook_scale=0.3, gain=2.0)
img, gt = make_batch(1234 + SEED, B, T, H, W)
res = {}
for name, sim in (("fp32", False), ("bf16sim", True)):
    p = O.to_torch_state(sd)
    res[name] = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p, bf16sim=sim)
eng = VQVAEEngine(sd, "cuda:0", dtype="bf16")
x = torch.from_numpy(img).reshape(B * T, 6, H, W).cuda()
y = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
recon, diff, S = eng.loss_and_backward(x, y, T=T)
torch.cuda.synchronize()
dec = ops.nhwc_to_nchw(S["dec"], 6).cpu()
for name in ("bf16sim", "fp32"):
    r = res[name]
    print(f"--- engine(bf16) vs oracle({name})")
    print("recon", recon.item(), r["recon"].item(), "latent", diff.item(), r["latent"].item())
    ref = r["fw"]["dec"].detach()
    print("dec rel L2", ((dec - ref).norm() / ref.norm()).item(), "max/scale", ((dec - ref).abs().max() / ref.abs().max()).item())
    for lvl in "tb":
        bad = (S["id_" + lvl].cpu() != r["fw"]["id_" + lvl]).reshape(-1)
        margin = O.vq_margin(r["fw"][f"q{lvl}_in"].detach(), torch.from_numpy(sd[f"quantize_{lvl}.embed"]))
        print(f"id_{lvl}: {int(bad.sum())} of {bad.numel()} differ; largest oracle margin at a mismatch", margin[bad].max().item() if bad.any() else 0.0,
              "median margin", margin.median().item())
    errs = sorted(((((eng.grads[k].cpu() - g).norm() / (g.norm() + 1e-30)).item(), k) for k, g in r["grads"].items()), reverse=True)
    print("grad rel L2 worst 5:", [(f"{e:.3e}", k) for e, k in errs[:5]], "median %.3e" % errs[len(errs) // 2][0])
a, b = res["bf16sim"], res["fp32"]
errs = sorted(((((a["grads"][k] - g).norm() / (g.norm() + 1e-30)).item(), k) for k, g in b["grads"].items()), reverse=True)
print("--- oracle(bf16sim) vs oracle(fp32): grad rel L2 worst 3", [(f"{e:.3e}", k) for e, k in errs[:3]], "median %.3e" % errs[len(errs) // 2][0])
# intermediates against the bf16-simulated oracle
fw = res["bf16sim"]["fw"]
def cmp(name, got_nhwc, ref_nchw):
    g = got_nhwc.float().cpu().permute(0, 3, 1, 2)
    r = ref_nchw.detach()
    print(f"  {name:12s} rel L2 {((g - r).norm() / r.norm()).item():.3e}  max/scale {((g - r).abs().max() / r.abs().max()).item():.3e}")
cmp("enc_b", S["eb"], fw["enc_b"]); cmp("enc_t", S["et"], fw["enc_t"])
cmp("enc_b_conv", S["cat_b"][..., 64:192], fw["enc_b_conv"]); cmp("enc_t_conv", S["d3"], fw["enc_t_conv"])
cmp("qt_in", S["qt_in"], fw["qt_in"].permute(0, 3, 1, 2)); cmp("quant_t", S["quant_t"], fw["quant_t"])
cmp("dec_t", S["cat_b"][..., 0:64], torch.zeros(1) + 0 if "dec_t" not in fw else fw["dec_t"]) if "dec_t" in fw else None
cmp("qb_in", S["qb_in"], fw["qb_in"].permute(0, 3, 1, 2)); cmp("quant_b", S["cat_d"][..., 64:128], fw["quant_b"])
