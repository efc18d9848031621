"""FREE-RUNNING figures at the benched sizes, anchored on the oracle's own fp64 evaluation (VERDICT r05 'weak' 1, next-round item 1b).

The branch-forced tests (tests/test_gan_gpu.py, tests/test_timed_size_oracle_gpu.py) compare arithmetic: with the code indices and the ReLU / LeakyReLU
branches of the engine forced onto the oracle, gradients agree to 1e-5 .. 4e-5.  What a user gets is the free-running step, where a pre-activation
within rounding of zero may take the other branch and move a gradient by that unit's whole contribution -- and that is as true of the REFERENCE's fp32
arithmetic as of the engine's: the fp32 oracle (torch-CPU, the reference's own kernels) is itself off its exact value by more than 1e-3 on some of
these gradients.  So the yardstick here is the same restatement evaluated in float64 (oracle.to_torch_state(dtype=torch.float64): every function takes
its dtype from its arguments), and

        err(engine, fp64)  <=  max(FLOOR, K * err(fp32 oracle, fp64))

-- for config 3 per tensor (bf16 rounding flips are so many that every tensor sees their average); for config 5 on the worst tensor and on the median
tensor (fp32: a handful of near-tie units, each of which moves ONE tensor by its whole contribution in whichever implementation happens to flip it -- a
per-tensor ratio compares two draws of a lottery; round 6 measured ratios up to 12 on single tensors with the two distributions within 2x of each other).
K = 2 where the engine's forward is as accurate as torch's direct convolutions (the direct kernels, 1e-6 of scale; the discriminators); K = 4.5 for the
default generator engine: its Winograd F(4x4) forwards are 2e-5 of scale, the number of near-tie ReLU units that flip grows with the forward
error, a gradient's error with the square root of the flips (contributions of random sign): sqrt(2e-5 / 1e-6) = 4.5.  Nothing is forced on the engine's side; the oracle runs take the ENGINE's code indices (index near-ties are gated on the oracle's fp64 top-2 margin by the
tests named above -- a flipped code is an O(1) local change of the function, not a rounding effect).  The figures are printed."""
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, make_disc_state, make_vgg_lpips_state

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _rel_scale(got, want):
    want = want.double()
    scale = max(want.abs().max().item(), want.pow(2).mean().sqrt().item()) + 1e-300
    return float((got.double().cpu() - want).abs().max().item() / scale)


def _rel_l2(got, want):
    want = want.double()
    return float((got.double().cpu() - want).norm() / (want.norm() + 1e-300))


def _threads():
    """CPU threads of the oracle runs: at most 16 (the pool's cgroup quota), so that the fp32 oracle's own summation order -- and with it WHICH near-tie
    units it flips -- is the same wherever the quota allows 16 (oneDNN partitions reductions by thread count)."""
    from _fullsize_oracle import cgroup_cpus
    return min(cgroup_cpus(), 16)


def _report(tag, e_eng, e_32, floor, k):
    names = sorted(e_eng, key=lambda n: -e_eng[n])
    bad = [(n, e_eng[n], e_32[n]) for n in names if e_eng[n] > max(floor, k * e_32[n])]
    med = lambda d: sorted(d.values())[len(d) // 2]
    print(f"[fp64 anchor, {tag}] engine vs fp64: worst {e_eng[names[0]]:.2e} ({names[0]}), median {med(e_eng):.2e}; fp32 oracle vs fp64: worst "
          f"{max(e_32.values()):.2e} ({max(e_32, key=e_32.get)}), median {med(e_32):.2e}; largest engine/fp32-oracle ratio among tensors over the floor "
          f"{max([e_eng[n] / max(e_32[n], 1e-30) for n in names if e_eng[n] > floor], default=0.0):.1f}; outside max({floor:g}, {k:g} x): {bad[:4]}")
    return bad


N5, H5, W5, WIN5 = 30, 256, 256, 16          # bench.py's c5 leg: one 30-frame clip, 16-frame window


def _c5_oracle(sd, sd3, sd2, img, gt, ids, c, gen_iter, dtype):
    from oracle import disc_oracle as D
    from oracle import faceoff_oracle as O
    p = O.to_torch_state(sd, dtype=dtype)
    p3, p2 = D.to_torch_state(sd3, dtype=dtype), D.to_torch_state(sd2, dtype=dtype)
    gtt = torch.from_numpy(gt).reshape(N5, 3, H5, W5).to(dtype)
    x = torch.from_numpy(img).to(dtype)
    with torch.set_grad_enabled(gen_iter):
        fw = O.vqvae_forward(x, p, training=True, force_ids=ids)
    out = fw["dec"][:, :3]
    r = c["random_idx"]
    x_fake, x_real = out[r:r + WIN5].unsqueeze(0), gtt[r:r + WIN5].unsqueeze(0)
    if gen_iter:
        recon, latent = torch.nn.functional.mse_loss(out, gtt), fw["diff"].mean()
        g2d, g3d = D.generator_gan_losses(x_fake, x_real, p3, p2, c["frame_id"], c["flip_real"], c["flip_fake"], {}, {})
        (recon + latent + g2d + g3d).backward()
        return {k: v.grad for k, v in p.items() if v.requires_grad}, dict(recon=recon.item(), latent=latent.item(), g_loss_2d=g2d.item(), g_loss_3d=g3d.item())
    dl3, dl2 = D.discriminator_losses(x_fake, x_real, p3, p2, c["frame_id"], c["flip_real"], c["flip_fake"], {}, {})
    dl3.backward()
    dl2.backward()
    g = {"d3." + k: v.grad for k, v in p3.items() if v.requires_grad}
    g.update({"d2." + k: v.grad for k, v in p2.items() if v.requires_grad})
    return g, dict(d_loss_3d=dl3.item(), d_loss_2d=dl2.item())


@pytest.mark.parametrize("gen_iter", [True, False], ids=["generator", "discriminator"])
def test_c5_free_running_at_the_benched_size_vs_fp64(gen_iter, monkeypatch):
    """Config 5 as bench.py's c5 leg times it, NOTHING forced on the engine: the generator iteration's 70 generator gradients (default engine, and the
    same iteration on the direct kernels) / the discriminator iteration's gradients of both discriminators against the fp64 evaluation, worst and median
    tensor bounded by the fp32 oracle's own distance to it (module docstring).  Round 6 recorded (generator): default engine worst 5.8e-3 / median
    2.0e-3, fp32 oracle 2.9e-3 / 6.1e-4; (discriminator) engine 3.8e-3 / 2.3e-4, fp32 oracle 4.8e-3 / 1.7e-4."""
    from faceoff_amd.disc import DiscEngine
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.gan_trainer import GANTrainer
    sd, sd3, sd2 = make_state_dict(0, codebook_scale=0.3, gain=2.0), make_disc_state(1, 3), make_disc_state(2, 2)
    img, gt = make_batch(55, 1, N5, H5, W5)
    x_img = torch.from_numpy(img).reshape(N5, 6, H5, W5).cuda()
    x_gt = torch.from_numpy(gt).reshape(N5, 3, H5, W5).cuda()
    c = dict(random_idx=5, frame_id=7, flip_real=True, flip_fake=False) if gen_iter else dict(random_idx=11, frame_id=3, flip_real=False, flip_fake=True)
    runs = {}
    for mode in (("default", "direct kernels") if gen_iter else ("default",)):  # (the discriminator iteration only runs the generator forward)
        if mode == "direct kernels":
            monkeypatch.setenv("FACEOFF_NO_WINOGRAD", "1")
        eng = VQVAEEngine(sd, "cuda:0")
        assert eng.winograd == (mode == "default")
        d3, d2 = DiscEngine(sd3, "cuda:0", dims=3, n_frames=WIN5 - 1), DiscEngine(sd2, "cuda:0", dims=2)
        tr = GANTrainer(eng, d3, d2, lr=3e-4, d_lr=1e-4, window=WIN5)
        tr.optimizer.step = lambda grad_scale=1.0: None                   # keep the gradients, skip the updates
        d3.adam_step = lambda *a_, **k_: None
        d2.adam_step = lambda *a_, **k_: None
        if not gen_iter:
            tr.iteration = 1
        # free-running: the engine's own codes, its own branches (the direct-kernel engine on the default engine's codes, so that ONE oracle
        # evaluation serves both: where the two would choose differently is an index near-tie, gated elsewhere)
        o = tr.step(x_img, x_gt, c, force_ids=None if mode == "default" else tuple(t.cuda() for t in runs["default"][2]))
        torch.cuda.synchronize()
        if gen_iter:
            got = {k: v.cpu() for k, v in eng.grads.items()}
        else:
            got = {"d3." + k: v.cpu() for k, v in d3.grads.items()}
            got.update({"d2." + k: v.cpu() for k, v in d2.grads.items()})
        runs[mode] = (got, {k: v.item() for k, v in o.items()}, tuple(t.cpu() for t in tr.last_ids))
        del eng, d3, d2, tr
        torch.cuda.empty_cache()
    ids = runs["default"][2]
    prev = torch.get_num_threads()
    torch.set_num_threads(_threads())
    try:
        g32, l32 = _c5_oracle(sd, sd3, sd2, img, gt, ids, c, gen_iter, torch.float32)
        g64, l64 = _c5_oracle(sd, sd3, sd2, img, gt, ids, c, gen_iter, F64)
    finally:
        torch.set_num_threads(prev)
    tot = max(v.abs().max().item() for v in g64.values())
    keep = [k for k, v in g64.items() if v.abs().max().item() >= 1e-4 * tot]        # (a bias in front of an InstanceNorm: its gradient is zero up to rounding)
    e_32 = {k: _rel_scale(g32[k], g64[k]) for k in keep}
    med = lambda d: sorted(d.values())[len(d) // 2]
    for mode, (got, losses, _) in runs.items():
        for k, v in l64.items():
            np.testing.assert_allclose(losses[k], v, rtol=1e-3, err_msg=f"{mode}: {k}")
        e_eng = {k: _rel_scale(got[k], g64[k]) for k in keep}
        K = C5_K_WINOGRAD if (gen_iter and mode == "default") else C5_K
        _report(f"C5 {'generator' if gen_iter else 'discriminator'} iteration, 30 x 256 x 256, window 16, free-running, {mode} engine", e_eng, e_32, C5_FLOOR, K)
        assert max(e_eng.values()) <= max(C5_FLOOR, K * max(e_32.values())), (mode, max(e_eng.items(), key=lambda kv: kv[1]), max(e_32.values()))
        assert med(e_eng) <= max(C5_FLOOR, K * med(e_32)), (mode, med(e_eng), med(e_32))
        assert max(e_eng.values()) <= C5_CAP, (mode, max(e_eng.items(), key=lambda kv: kv[1]))


C5_FLOOR, C5_K, C5_K_WINOGRAD, C5_CAP = 1e-3, 2.0, 4.5, 1e-2


B3, T3 = 2, 5                                 # two of config 3's 32 clips: the fp64 evaluation of the LPIPS branch is ~20 s of host time per clip


def test_c3_free_running_branches_vs_fp64_accumulation():
    """Config 3 (bf16 operands): the engine is ONE MORE summation order of the bf16 policy the oracle's bf16sim states (same rounding points; what differs
    between two implementations is which way a stored value that sits on a bf16 rounding boundary goes, and through it ReLU / max-pool branches).  The
    yardstick: the bf16-simulated oracle with float64 accumulation between the same rounding points; the bound: the fp32-accumulating bf16-simulated
    oracle's own distance to it.  Codes teacher-forced onto the fp32 oracle's for all three (as every bf16 parity test), branches free."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    from oracle import faceoff_oracle as O
    H, W = 256, 256
    sd, lp = make_state_dict(0, codebook_scale=0.3, gain=2.0), make_vgg_lpips_state(7)
    g = torch.Generator().manual_seed(2026)
    img = torch.rand((B3, T3, 6, H, W), generator=g) * 2 - 1
    gt = torch.rand((B3, T3, 3, H, W), generator=g) * 2 - 1
    prev = torch.get_num_threads()
    torch.set_num_threads(_threads())
    try:
        res = {}
        ids = None
        for dt in (torch.float32, F64):
            p = O.to_torch_state(sd, dtype=dt)
            lpt = {k: torch.as_tensor(v).to(dt) for k, v in lp.items()}
            r = O.run_step(img.to(dt), gt.to(dt), p, lpt, training=True, bf16sim=True, lpips_bf16sim=True, force_ids=ids)
            r["loss"].backward()
            if ids is None:
                ids = (r["fw"]["id_t"], r["fw"]["id_b"])
            res[dt] = ({k: v.grad for k, v in p.items() if v.requires_grad}, {k: float(r[k]) for k in ("recon", "latent", "perceptual")})
            del r
    finally:
        torch.set_num_threads(prev)
    eng = VQVAEEngine(sd, "cuda:0", dtype="bf16")
    tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(lp, dtype="bf16").cuda())
    tr.optimizer.step = lambda grad_scale=1.0: None
    N = B3 * T3
    recon, latent, perceptual = tr.step(img.reshape(N, 6, H, W).cuda(), gt.reshape(N, 3, H, W).cuda(), T=T3, force_ids=tuple(t.cuda() for t in ids))
    torch.cuda.synchronize()
    g64, l64 = res[F64]
    np.testing.assert_allclose([recon.item(), latent.item(), perceptual.item()], [l64["recon"], l64["latent"], l64["perceptual"]], rtol=5e-3)
    e_eng = {k: _rel_l2(eng.grads[k], g64[k]) for k in g64}
    e_32 = {k: _rel_l2(res[torch.float32][0][k], g64[k]) for k in g64}
    # (round 6 recorded: engine worst 9.0e-3 / median 3.8e-4, fp32-accumulating oracle 8.8e-3 / 3.5e-4, largest per-tensor ratio 1.2)
    bad = _report(f"C3 bf16, {B3} x {T3} x 256 x 256, rel-L2, codes forced, branches free", e_eng, e_32, C3_FLOOR, C3_K)
    assert not bad, bad[:6]
    assert max(e_eng.values()) <= C3_CAP, max(e_eng.items(), key=lambda kv: kv[1])


C3_FLOOR, C3_K, C3_CAP = 1e-3, 2.0, 2e-2
