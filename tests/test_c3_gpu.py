"""BASELINE config 3 as a WHOLE training step on the MI355X: VQ-VAE-2 (fp32) + LPIPS/VGG-16 perceptual loss in bf16
(train_faceoff_perceptual.py:32-47,92-107 with loss.VQLPIPS, loss.py:27-33): recon + latent + perceptual, backward through
both, all 70 gradients.  1e-3 against an fp32 oracle is not attainable with bf16 operands, so the checker is
the oracle with the SAME rounding points in its LPIPS branch (every stored activation, its gradient and every VGG filter
rounded to bfloat16, fp32 accumulation; oracle.run_step(lpips_bf16sim=True)); the deviation from the pure-fp32 oracle is
bounded beside it.  Plus size-independent properties of the step at the full config-3 size (160 frames of 256x256)."""
import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, make_vgg_lpips_state

pytestmark = pytest.mark.gpu


def _rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def test_c3_whole_step_bf16_lpips_vs_bf16_simulated_oracle():
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    from oracle import faceoff_oracle as O
    B, T, H, W = 2, 2, 64, 64
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    lp = make_vgg_lpips_state(7)
    img, gt = make_batch(1234, B, T, H, W)
    lpt = {k: torch.from_numpy(v) for k, v in lp.items()}
    ref = {}
    for sim in (True, False):
        ref[sim] = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), O.to_torch_state(sd), lpips_state=lpt, lpips_bf16sim=sim)
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(lp, dtype="bf16").cuda())
    tr.optimizer.step = lambda grad_scale=1.0: None          # keep the gradients, skip the update
    recon, latent, perceptual = tr.step(torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda())
    torch.cuda.synchronize()
    r = ref[True]
    np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=1e-3)          # the VQ-VAE itself stays fp32
    np.testing.assert_allclose(latent.item(), r["latent"].item(), rtol=1e-3)
    np.testing.assert_allclose(perceptual.item(), r["perceptual"].item(), rtol=2e-3)
    np.testing.assert_allclose(perceptual.item(), ref[False]["perceptual"].item(), rtol=2e-2)
    worst_sim, worst_f32 = (0.0, ""), (0.0, "")
    for n, gref in r["grads"].items():
        got = eng.grads[n].cpu().numpy()
        e_sim, e_f32 = _rel_l2(got, gref.numpy()), _rel_l2(got, ref[False]["grads"][n].numpy())
        worst_sim, worst_f32 = max(worst_sim, (e_sim, n)), max(worst_f32, (e_f32, n))
        assert e_sim <= 3e-2, (n, e_sim)
        assert e_f32 <= 1e-1, (n, e_f32)
    sim_vs_f32 = max(_rel_l2(ref[True]["grads"][n].numpy(), ref[False]["grads"][n].numpy()) for n in r["grads"])
    print(f"[C3 step] perceptual {perceptual.item():.6f} (bf16-sim oracle {r['perceptual'].item():.6f}, fp32 oracle "
          f"{ref[False]['perceptual'].item():.6f}); worst gradient rel-L2 vs bf16-sim {worst_sim}, vs fp32 {worst_f32}; "
          f"bf16-sim oracle vs fp32 oracle {sim_vs_f32:.3e}")


def test_c3_full_size_step_is_finite_reproducible_and_chunk_invariant():
    """160 frames of 256x256, T=5 (the size bench.py's `c3` leg times): finite losses and gradients, every parameter
    tensor receives a gradient, two runs give the same gradient bits, and running the LPIPS branch in frame chunks (the
    2 GiB-window path) changes nothing beyond the order of the loss atomics."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    N, T = 160, 5
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.rand((N, 6, 256, 256), device="cuda", generator=g) * 2 - 1
    gt = torch.rand((N, 3, 256, 256), device="cuda", generator=g) * 2 - 1
    lp = make_vgg_lpips_state(7)
    runs = []
    for chunked in (False, False, True):
        eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0")
        vq = VQLPIPS(lp, dtype="bf16").cuda()
        tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=vq)
        tr.optimizer.step = lambda grad_scale=1.0: None
        if chunked:
            lpeng = vq._bind(torch.device("cuda:0"))
            lpeng.window_bytes = 64 * 256 * 256 * 64 * 2          # 64 frames per pass -> chunks of 54, 53, 53
            assert lpeng.max_frames(256, 256) < N
        recon, latent, perceptual = tr.step(img, gt, T=T)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in (recon, latent, perceptual)) and perceptual.item() > 0
        assert torch.isfinite(eng.flat_grads).all()
        for key, (off, n) in eng.offsets.items():
            assert eng.flat_grads[off:off + n].abs().max().item() > 0, key
        runs.append((eng.flat_grads.clone(), perceptual.item(), recon.item()))
        del eng, tr, vq
        torch.cuda.empty_cache()
    assert torch.equal(runs[0][0], runs[1][0]), "the C3 step's gradients are not reproducible"
    np.testing.assert_allclose(runs[1][1:], runs[0][1:], rtol=1e-5)
    np.testing.assert_allclose(runs[2][1:], runs[0][1:], rtol=1e-5)
    d = (runs[2][0] - runs[0][0]).abs().max().item()
    assert d <= 1e-5 * runs[0][0].abs().max().item(), d
