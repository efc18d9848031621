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
from _observed import Observed

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
        assert e_sim <= 5e-4, (n, e_sim)          # observed 1.6e-4 at worst: the perceptual gradient is a small part of these sums (the
        assert e_f32 <= 2e-3, (n, e_f32)          # undiluted statement is test_c3_perceptual_only_gradient below); vs fp32: observed 3.7e-4
    sim_vs_f32 = max(_rel_l2(ref[True]["grads"][n].numpy(), ref[False]["grads"][n].numpy()) for n in r["grads"])
    print(f"[C3 step] perceptual {perceptual.item():.6f} (bf16-sim oracle {r['perceptual'].item():.6f}, fp32 oracle "
          f"{ref[False]['perceptual'].item():.6f}); worst gradient rel-L2 vs bf16-sim {worst_sim}, vs fp32 {worst_f32}; "
          f"bf16-sim oracle vs fp32 oracle {sim_vs_f32:.3e}")


def test_c3_perceptual_only_gradient_through_lpips_and_the_engine():
    """The LPIPS backward on its own: loss = 0 * recon + 0 * latent + 1 * perceptual, so nothing dilutes it.  g_dec from
    the bf16 LPIPS branch alone against d perceptual / d dec of the oracle with the same rounding points, then that gradient through the fp32
    engine's backward: all 70 parameter gradients against the oracle's, each within twice its recorded error."""
    from faceoff_amd import ops
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from oracle import faceoff_oracle as O
    B, T, H, W = 2, 2, 64, 64
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    lp = make_vgg_lpips_state(7)
    img, gt = make_batch(1234, B, T, H, W)
    lpt = {k: torch.from_numpy(v) for k, v in lp.items()}
    p = O.to_torch_state(sd)
    r = O.run_step(torch.from_numpy(img), torch.from_numpy(gt), p, lpips_state=lpt, lpips_bf16sim=True, weights=(0.0, 0.0, 1.0))
    r["fw"]["dec"].retain_grad()
    r["loss"].backward()
    g_dec_ref = r["fw"]["dec"].grad[:, :3]
    eng = VQVAEEngine(sd, "cuda:0")
    vq = VQLPIPS(lp, dtype="bf16").cuda()
    x = torch.from_numpy(img).reshape(B * T, 6, H, W).cuda()
    y = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
    S = eng.forward(x, training=True, T=T)
    g_dec = torch.zeros_like(S["dec"])
    perceptual = vq.loss_and_grad(y, S["dec"], g_dec, 1.0)
    eng.backward(S, g_dec, torch.zeros(1, device="cuda"))
    torch.cuda.synchronize()
    obs = Observed("c3_perceptual_only_2x2x64x64")
    np.testing.assert_allclose(perceptual.item(), r["perceptual"].item(), rtol=2e-3)
    got = ops.nhwc_to_nchw(g_dec, 6).cpu().numpy()
    assert np.abs(got[:, 3:]).max() == 0.0                           # channels 3-5 of dec are not in the loss (:37)
    e_dec = _rel_l2(got[:, :3], g_dec_ref.numpy())
    obs.check("g_dec", e_dec, cap=3e-2)
    errs = sorted(((_rel_l2(eng.grads[n].cpu().numpy(), p[n].grad.numpy()), n) for n in eng.grads), reverse=True)
    print(f"[C3 perceptual-only] d perceptual / d dec rel-L2 vs bf16-sim oracle {e_dec:.3e}; parameter gradients: worst {errs[0]}, median {errs[len(errs) // 2][0]:.3e}")
    for e, n in errs:
        obs.check("grad:" + n, e, cap=6e-2)
    obs.flush()


C3_FIXTURES = [(2, 2, 64, 64, 0), (3, 3, 48, 80, 11)]       # (LPIPS pools four times: multiples of 16; 3 x 5 at the deepest tap)


@pytest.mark.parametrize("B,T,H,W,seed", C3_FIXTURES)
def test_c3_as_timed_bf16_engine_plus_bf16_lpips_vs_bf16_simulated_oracle(B, T, H, W, seed):
    """Config 3 AS bench.py TIMES IT: VQVAEEngine(dtype="bf16") + VQLPIPS(dtype="bf16") in ONE FaceOffTrainer.step --
    the LPIPS target branch and four of the five heads on side streams, the bf16 g_dec hand-over -- against
    oracle.train_step(bf16sim=True, lpips_bf16sim=True).  Twice: free-running (losses, margin-gated code indices) and teacher-forced onto the
    oracle's codes (flip-free: decoder output and all 70 gradients, each within twice its recorded error)."""
    from faceoff_amd import ops
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    from oracle import faceoff_oracle as O
    sd = make_state_dict(seed, codebook_scale=0.3, gain=2.0)
    lp = make_vgg_lpips_state(7)
    img, gt = make_batch(1234 + seed, B, T, H, W)
    lpt = {k: torch.from_numpy(v) for k, v in lp.items()}
    r = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), O.to_torch_state(sd), lpips_state=lpt, lpips_bf16sim=True, bf16sim=True)
    ids = (r["fw"]["id_t"], r["fw"]["id_b"])
    obs = Observed(f"c3_as_timed_{B}x{T}x{H}x{W}")
    for forced in (False, True):
        eng = VQVAEEngine(sd, "cuda:0", dtype="bf16")
        tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(lp, dtype="bf16").cuda())
        assert tr.lpips_stream is not None                       # the overlapped form, as timed
        tr.optimizer.step = lambda grad_scale=1.0: None          # keep the gradients, skip the update
        recon, latent, perceptual = tr.step(torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda(), force_ids=ids if forced else None)
        torch.cuda.synchronize()
        np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=2e-3 if forced else 2e-2)
        np.testing.assert_allclose(latent.item(), r["latent"].item(), rtol=2e-3 if forced else 2e-2)
        np.testing.assert_allclose(perceptual.item(), r["perceptual"].item(), rtol=5e-3 if forced else 5e-2)
        id_t, id_b = (t.cpu() for t in tr.last_ids)
        if not forced:
            bad_t = (id_t != ids[0]).reshape(-1)
            m = O.vq_margin(r["fw"]["qt_in"].detach(), torch.from_numpy(sd["quantize_t.embed"]))
            assert bool((m[bad_t] < 1e-2).all()) and bad_t.float().mean().item() < 1e-2
            agree_b = (id_b == ids[1]).float().mean().item()
            assert agree_b > (0.99 if not bad_t.any() else 0.5), agree_b
            print(f"[C3 as timed {B}x{T}x{H}x{W}, free-running] losses {recon.item():.5f} / {latent.item():.5f} / {perceptual.item():.5f} (oracle "
                  f"{r['recon'].item():.5f} / {r['latent'].item():.5f} / {r['perceptual'].item():.5f}); top flips {int(bad_t.sum())}, bottom agreement {agree_b:.4f}")
            continue
        assert torch.equal(id_t, ids[0]) and torch.equal(id_b, ids[1])
        errs = sorted(((_rel_l2(eng.grads[n].cpu().numpy(), g.numpy()), n) for n, g in r["grads"].items()), reverse=True)
        print(f"[C3 as timed {B}x{T}x{H}x{W}, teacher-forced] perceptual {perceptual.item():.6f} (oracle {r['perceptual'].item():.6f}); gradients vs the "
              f"bf16-simulated oracle: worst {errs[0]}, median {errs[len(errs) // 2][0]:.3e}")
        for e, n in errs:
            obs.check("grad:" + n, e, cap=8e-2)
    obs.flush()


def test_c3_as_timed_full_size_is_finite_reproducible_and_reaches_every_parameter():
    """160 frames of 256x256, T = 5, bf16 engine + bf16 LPIPS in one step (the `c3` leg of bench.py): finite losses and gradients, every
    parameter tensor receives a gradient, and two runs give the same gradient AND loss bits (no float atomics left in the loss sums)."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    N, T = 160, 5
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.rand((N, 6, 256, 256), device="cuda", generator=g) * 2 - 1
    gt = torch.rand((N, 3, 256, 256), device="cuda", generator=g) * 2 - 1
    lp = make_vgg_lpips_state(7)
    runs = []
    for rep in range(2):
        eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), "cuda:0", dtype="bf16")
        tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(lp, dtype="bf16").cuda())
        tr.optimizer.step = lambda grad_scale=1.0: None
        recon, latent, perceptual = tr.step(img, gt, T=T)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in (recon, latent, perceptual)) and perceptual.item() > 0
        assert torch.isfinite(eng.flat_grads).all()
        for key, (off, n) in eng.offsets.items():
            assert eng.flat_grads[off:off + n].abs().max().item() > 0, key
        runs.append((eng.flat_grads.clone(), recon.item(), latent.item(), perceptual.item()))
        del eng, tr
        torch.cuda.empty_cache()
    assert torch.equal(runs[0][0], runs[1][0]), "the C3 step's gradients are not reproducible"
    assert runs[0][1:] == runs[1][1:], (runs[0][1:], runs[1][1:])


def test_c3_full_size_step_is_finite_reproducible_and_chunk_invariant():
    """160 frames of 256x256, T=5 (the size bench.py's `c3` leg times): finite losses and gradients, every parameter
    tensor receives a gradient, two runs give the same gradient bits, and running the LPIPS branch in frame chunks (the
    2 GiB-window path) changes nothing beyond the rounding of the loss sums."""
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
    assert runs[1][1:] == runs[0][1:], (runs[1][1:], runs[0][1:])        # same launches: the loss scalars are the same bits (no float atomics)
    np.testing.assert_allclose(runs[2][1:], runs[0][1:], rtol=1e-5)       # frame chunks: other partial sums, other rounding
    d = (runs[2][0] - runs[0][0]).abs().max().item()
    assert d <= 1e-5 * runs[0][0].abs().max().item(), d
