"""Whole-step parity on a real MI355X: forward + backward (+ Adam) of the HIP engine against
(1) golden outputs of the reference itself (tests/golden/*.npz) and (2) the CPU oracle on the same
seeded inputs.  Tolerance 1e-3 relative to each tensor's scale; VQ indices bit-exact (mismatches
tolerated only where the reference's own top-2 margin is below the upstream fp32 error)."""
import os

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, golden_state

pytestmark = pytest.mark.gpu
SUB = 61


def _sub(t):
    return t.detach().reshape(-1)[::SUB].cpu().numpy()


def _stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.pow(2).sum().item(), t.abs().max().item()])


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def _engine_step(g, T=None, forced=False):
    """forced: the engine runs on the REFERENCE's code indices (VQVAEEngine.forward(force_ids=...): the search is skipped)"""
    from faceoff_amd.engine import VQVAEEngine
    B, T_, H, W = (int(g[k]) for k in "BTHW")
    sd = golden_state(g)
    eng = VQVAEEngine(sd, "cuda:0")
    img, gt = make_batch(int(g["seed_x"]), B, T_, H, W)
    img = torch.from_numpy(img).reshape(B * T_, 6, H, W).cuda()
    gt = torch.from_numpy(gt).reshape(B * T_, 3, H, W).cuda()
    force = None
    if forced:
        force = tuple(torch.from_numpy(g["id_" + l].astype(np.int64)).reshape(B * T_, H // s, W // s).cuda() for l, s in (("t", 8), ("b", 4)))
    recon, diff, S = eng.loss_and_backward(img, gt, T=T or T_, force_ids=force)
    torch.cuda.synchronize()
    return eng, recon, diff, S, img, gt


def _dilate(m, r):
    return torch.nn.functional.max_pool2d(m.float().unsqueeze(1), 2 * r + 1, 1, r).squeeze(1) > 0


def _outside_flipped_receptive_fields(bad_t, bad_b):
    """bool [N,H,W] at image resolution: the pixels of `dec` no flipped code can reach (derivation: tests/test_bf16_engine_gpu.py -- a top
    code enters through upsample_t's k4 s2 p1, then three 3x3 stages + one latent pixel per transposed stage: 5 latent pixels each side)."""
    up = _dilate(bad_t, 1).repeat_interleave(2, 1).repeat_interleave(2, 2)
    reach = _dilate(bad_b | up, 5)
    return ~reach.repeat_interleave(4, 1).repeat_interleave(4, 2)


def _check_against_golden(g, eng, recon, diff, S, literal=False, what="", expect_flips=None, tol0=1e-3):
    """Bounds: 1e-3 of each tensor's scale (BASELINE.json north_star), always.  An index may differ only where the REFERENCE's own top-2
    distance margin is below 1e-4 (an fp32 near-tie: any summation order may flip it).  A flipped code changes `dec` locally by O(1), so
    when that happens nothing is widened (round 4 had a 1e-1 fallback here): `dec` is held to the same bound OUTSIDE the flipped codes'
    receptive fields, and losses, all 70 gradients and the EMA buffers are checked on a second step that runs on the reference's codes
    (teacher-forced: the search is skipped, everything else is the same launches)."""
    from faceoff_amd import ops
    dec = ops.nhwc_to_nchw(S["dec"], 6)
    flips, bad = 0, {}
    for lvl in "tb":
        got = S["id_" + lvl].cpu().numpy().astype(np.int16).reshape(-1)
        b = got != g["id_" + lvl].reshape(-1)
        assert np.all(g["margin_" + lvl][b] < 1e-4), f"id_{lvl}: {int(b.sum())} mismatches outside the near-tie gate"
        assert b.mean() < 2e-3
        flips += int(b.sum())
        bad[lvl] = torch.from_numpy(b).reshape(S["id_" + lvl].shape)
    # the fixtures whose reference margins leave no near-tie (b1_literal, c1, c1w: smallest top-2 margin >> fp32 error) must
    # reproduce EVERY index: a flip there is a bug, not rounding
    if expect_flips is not None:
        assert flips == expect_flips, f"{what}: {flips} code-index flips, expected {expect_flips}"
    keep = _outside_flipped_receptive_fields(bad["t"], bad["b"]).unsqueeze(1).expand(-1, 6, -1, -1)
    assert keep.float().mean().item() > 0.5, "the flipped codes' receptive fields cover most of the fixture"
    if g["dec"].ndim == 4:
        got_dec, want_dec, keep = dec.cpu().numpy(), g["dec"], keep.numpy()
    else:
        got_dec, want_dec, keep = _sub(dec), g["dec"], keep.reshape(-1)[::SUB].numpy()
    obs = {"dec": float(np.abs(got_dec - want_dec)[keep].max() / (np.abs(want_dec).max() + 1e-30))}
    assert obs["dec"] < tol0, obs
    if flips:
        eng, recon, diff, S, *_ = _engine_step(g, forced=True)
        assert all(np.array_equal(S["id_" + lvl].cpu().numpy().reshape(-1), g["id_" + lvl].reshape(-1)) for lvl in "tb")
        dec = ops.nhwc_to_nchw(S["dec"], 6)
        obs["dec_forced"] = _rel(dec.cpu().numpy() if g["dec"].ndim == 4 else _sub(dec), g["dec"])
        assert obs["dec_forced"] < tol0, obs
    tol = tol0
    np.testing.assert_allclose(recon.item(), float(g["recon"]), rtol=1e-3)
    np.testing.assert_allclose(diff.item(), float(g["latent"]), rtol=1e-3)
    names = [str(n) for n in g["param_names"]]
    gs = np.stack([_stats(eng.grads[n]) for n in names])
    l2, want_l2 = np.sqrt(gs[:, 1]), np.sqrt(g["grad_stats"][:, 1])
    obs["grad_l2"] = float(np.max(np.abs(l2 - want_l2) / want_l2))
    off_l2 = [(n, a, b) for n, a, b in zip(names, l2, want_l2) if abs(a - b) > tol * abs(b)]
    assert not off_l2, f"gradient L2 norms off: {off_l2}"
    # every tensor's strided subsample against the tensor's own scale
    off, worst = 0, (0.0, "")
    for i, n in enumerate(names):
        got = _sub(eng.grads[n])
        want = g["grad_sub"][off:off + len(got)]
        off += len(got)
        scale = max(np.sqrt(g["grad_stats"][i, 1] / eng.grads[n].numel()), np.abs(want).max()) + 1e-30
        err = np.abs(got - want).max() / scale
        worst = max(worst, (float(err), n))
        assert err <= tol, (n, err)
    obs["grad_sub"] = worst
    for n in names:
        if "grad_full." + n in g.files:
            assert _rel(eng.grads[n].cpu().numpy(), g["grad_full." + n]) < tol0, n
    for k, b in eng.buffers.items():
        np.testing.assert_allclose(_stats(b)[1], g["buf_stats." + k][1], rtol=1e-3)
    print(f"[parity {what}] index flips {flips}{' (gradients / buffers from the step on the reference codes)' if flips else ''}; observed max rel err: {obs}")
    return flips, obs


def test_c1_e2e_vs_reference_golden(golden_dir):
    """BASELINE config 1 (64x64, T=2, bs=2): forward, losses, indices, all 70 gradients, EMA buffers."""
    g = np.load(os.path.join(golden_dir, "c1_e2e.npz"))
    eng, recon, diff, S, img, gt = _engine_step(g)
    _check_against_golden(g, eng, recon, diff, S, what="c1", expect_flips=0)
    # Adam step (train_faceoff_perceptual.py:107) then an eval forward pins the whole state update
    from faceoff_amd import ops
    m, v = torch.zeros_like(eng.flat_params), torch.zeros_like(eng.flat_params)
    ops.adam_flat(eng.flat_params, eng.flat_grads, m, v, 3e-4, 1)
    names = [str(n) for n in g["param_names"]]
    after = np.concatenate([_sub(eng.params[n]) for n in names])
    np.testing.assert_allclose(after, g["param_after_sub"], rtol=1e-3, atol=3e-5)
    S2 = eng.forward(img, training=False, T=int(g["T"]))
    dec2 = ops.nhwc_to_nchw(S2["dec"], 6)
    np.testing.assert_allclose(S2["diff"].item(), float(g["diff2"].reshape(-1)[0]), rtol=5e-2)
    assert np.abs(_sub(dec2) - g["dec2_sub"]).max() < 5e-2 * np.abs(g["dec2_sub"]).max()


def test_c1w_e2e_many_codes_vs_reference_golden(golden_dir):
    """96x96, codebooks centred on the latents: > 100 distinct codes in use on both levels."""
    g = np.load(os.path.join(golden_dir, "c1w_e2e.npz"))
    assert len(np.unique(g["id_t"])) > 100 and len(np.unique(g["id_b"])) > 100
    eng, recon, diff, S, img, gt = _engine_step(g)
    _check_against_golden(g, eng, recon, diff, S, what="c1w", expect_flips=0)


def test_b1_literal_reference_forward(golden_dir):
    """One clip of 4 frames (T = N): equals the reference's VQVAE.forward itself."""
    g = np.load(os.path.join(golden_dir, "b1_literal.npz"))
    eng, recon, diff, S, img, gt = _engine_step(g)
    _check_against_golden(g, eng, recon, diff, S, literal=True, what="b1 literal", expect_flips=0)


def test_c2_oneclip_vs_reference_golden(golden_dir):
    """C2 shape (256x256, T=5), one clip: losses, indices (margin-gated), gradient norms."""
    g = np.load(os.path.join(golden_dir, "c2_oneclip.npz"))
    eng, recon, diff, S, img, gt = _engine_step(g)
    _check_against_golden(g, eng, recon, diff, S, what="c2 one clip")


def test_c2_oneclip_direct_engine_is_tight_and_the_default_engine_differs_by_relu_near_ties_only(golden_dir, monkeypatch):
    """Where the 256x256 clip's first-layer filter gradient gets its 0.9e-3 (bound 1e-3) from -- measured (tools/probes/
    c2clip_probe.py): NOT from the arithmetic of any backward kernel (switching single data- / filter-gradient passes between
    their Winograd and direct forms moves it in the 4th digit) but from ~25 ReLU masks that the Winograd FORWARD passes'
    2e-5 of rounding flip on pre-activations within rounding of zero (enc_b.blocks.2 on F(4x4,2x2) alone accounts for half).  The inputs
    are noise, so that gradient is a sum of 82 000 terms of random sign (|sum| ~ 300 sigma) and one flipped mask moves it by a few sigma:
    any two fp32 implementations differ like this, the direct engine against torch-CPU by 2.5e-4.  So the margin is made explicit:
      (1) the engine on the direct kernels (5 such flips) must be within 5e-4 of the reference's golden on every tensor;
      (2) the default engine's saved activations equal the direct engine's to 1e-4 of scale, and every ReLU mask that differs does
          so on an element below 1e-4 of scale in both (a near-tie, gated exactly like the VQ near-ties); the count is bounded."""
    g = np.load(os.path.join(golden_dir, "c2_oneclip.npz"))
    e4, r4, d4, S4, *_ = _engine_step(g)
    assert e4.winograd and e4.winograd_max_tile == 4
    monkeypatch.setenv("FACEOFF_NO_WINOGRAD", "1")
    e2, r2, d2, S2, *_ = _engine_step(g)
    assert not e2.winograd
    flips2, obs2 = _check_against_golden(g, e2, r2, d2, S2, what="c2 one clip, direct kernels", tol0=5e-4)
    relu_flips, worst_at_flip = 0, 0.0
    for k, a in S4.items():
        b = S2.get(k)
        if torch.is_tensor(a) and torch.is_tensor(b) and a.is_floating_point() and a.shape == b.shape and a.dim() == 4:
            scale = b.abs().max().item() + 1e-30
            assert (a - b).abs().max().item() <= 1e-4 * scale, k
            diff_mask = (a > 0) != (b > 0)
            n = int(diff_mask.sum().item())
            if n:
                relu_flips += n
                worst_at_flip = max(worst_at_flip, torch.maximum(a.abs(), b.abs())[diff_mask].max().item() / scale)
    print(f"[c2 one clip] default (Winograd) vs direct engine: {relu_flips} ReLU-mask differences, largest |activation| at one "
          f"{worst_at_flip:.2e} of scale; direct engine vs golden: {obs2}")
    assert worst_at_flip <= 1e-4, worst_at_flip
    assert relu_flips <= 200, relu_flips


def test_e2e_vs_oracle_ragged():
    """Odd sizes the golden set does not hold: 3 clips of T=3 at 40x24 (tiles straddle frames, M tails)."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd import ops
    from oracle import faceoff_oracle as O
    B, T, H, W = 3, 3, 40, 24
    sd = make_state_dict(11, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(5, B, T, H, W)
    p = O.to_torch_state(sd)
    r = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p)
    eng = VQVAEEngine(sd, "cuda:0")
    x = torch.from_numpy(img).reshape(B * T, 6, H, W).cuda()
    y = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
    recon, diff, S = eng.loss_and_backward(x, y, T=T)
    dec = ops.nhwc_to_nchw(S["dec"], 6).cpu()
    np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=1e-3)
    np.testing.assert_allclose(diff.item(), r["latent"].item(), rtol=1e-3)
    flips = 0
    for lvl in "tb":                                          # indices: equal, or the oracle's own margin is a near-tie
        bad = (S["id_" + lvl].cpu() != r["fw"]["id_" + lvl]).reshape(-1)
        margin = O.vq_margin(r["fw"][f"q{lvl}_in"].detach(), torch.from_numpy(sd[f"quantize_{lvl}.embed"]))
        assert bool((margin[bad] < 1e-4).all()) and bad.float().mean().item() < 2e-3, (lvl, int(bad.sum()))
        flips += int(bad.sum())
    if flips:          # never a wider bound: the step again on the oracle's codes (teacher-forced), then everything at 1e-3
        eng = VQVAEEngine(sd, "cuda:0")
        recon, diff, S = eng.loss_and_backward(x, y, T=T, force_ids=(r["fw"]["id_t"].cuda(), r["fw"]["id_b"].cuda()))
        dec = ops.nhwc_to_nchw(S["dec"], 6).cpu()
    assert _rel(dec.numpy(), r["fw"]["dec"].detach().numpy()) < 1e-3
    tol, worst = 1e-3, (0.0, "")
    for n, gref in r["grads"].items():
        got = eng.grads[n].cpu().numpy()
        rms = gref.pow(2).mean().sqrt().item()
        err = np.abs(got - gref.numpy()).max() / max(rms, gref.abs().max().item())
        worst = max(worst, (float(err), n))
        assert err <= tol, (n, err)
    for k in eng.buffers:
        assert _rel(eng.buffers[k].cpu().numpy(), p[k].numpy()) < 1e-3, k
    print(f"[parity ragged] index flips {flips}; worst gradient rel err {worst}")


def test_perceptual_step_vs_oracle():
    """train_faceoff_perceptual.py:32-47,98-107 with the LPIPS term: recon + latent + perceptual, all gradients."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.loss import VQLPIPS
    from faceoff_amd.trainer import FaceOffTrainer
    from faceoff_amd.synth import make_vgg_lpips_state
    from oracle import faceoff_oracle as O
    B, T, H, W = 2, 2, 64, 64
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    lp = make_vgg_lpips_state(7)
    img, gt = make_batch(1234, B, T, H, W)
    p = O.to_torch_state(sd)
    r = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p, lpips_state={k: torch.from_numpy(v) for k, v in lp.items()})
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4, vqlpips=VQLPIPS(lp).cuda())
    tr.optimizer.step = lambda grad_scale=1.0: None          # keep the gradients, skip the update
    recon, latent, perceptual = tr.step(torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda())
    np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=1e-3)
    np.testing.assert_allclose(latent.item(), r["latent"].item(), rtol=1e-3)
    np.testing.assert_allclose(perceptual.item(), r["perceptual"].item(), rtol=1e-3)
    for n, gref in r["grads"].items():
        got = eng.grads[n].cpu().numpy()
        assert np.abs(got - gref.numpy()).max() <= 1e-3 * gref.abs().max().item(), n


def test_step_from_loader_batch_equals_step_on_concatenated_input():
    """process_data fused into the input-layout kernel (fo_nchw2_to_nhwc8): feeding the loader's 5-tuple gives
    identical losses and bit-identical gradients and updated parameters to feeding the concatenated [N,6,H,W] tensor."""
    from faceoff_amd import ops
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    g = torch.Generator().manual_seed(11)
    T, H, W = 3, 32, 48
    data = tuple(torch.rand((1, T, 3, H, W), generator=g) * 2 - 1 for _ in range(5))
    a, b = data[0][0].cuda(), data[2][0].cuda()
    assert torch.equal(ops.cat_nchw_to_nhwc8(a, b), ops.nchw_to_nhwc(torch.cat([a, b], 1), cpad=8))
    outs = []
    for mode in ("tuple", "cat"):
        eng = VQVAEEngine(make_state_dict(5, codebook_scale=0.3, gain=2.0), "cuda:0")
        tr = FaceOffTrainer(eng)
        if mode == "tuple":
            recon, latent, _, S = tr.step_from_batch(data)
            assert S == T
        else:
            recon, latent, _ = tr.step(torch.cat([a, b], 1), data[3][0].cuda(), T=T)
        torch.cuda.synchronize()
        outs.append((recon.item(), latent.item(), eng.flat_grads.clone(), eng.flat_params.clone()))
    np.testing.assert_allclose(outs[0][:2], outs[1][:2], rtol=1e-6)      # loss sums use float atomics: order varies
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])


def test_winograd_and_direct_conv3d_engines_agree():
    """The default engine (Conv3d / 3x3 128->128 layers as Winograd F(4x4 | 2x2, 3x3)) against the same engine on the
    direct kernels (FACEOFF_NO_WINOGRAD) at a size where both Winograd tiles are in play (128x128 frames, 8 frames).
    Forward: same code indices, outputs and losses to 1e-5.  Backward: the two forwards differ by ~1e-5 of scale, which
    moves a handful of pre-activations across zero, and a ReLU's derivative is discontinuous there (the same happens
    between any two fp32 implementations, the reference's included): every gradient tensor within the 1e-3 parity bound,
    and the median tensor within 1e-4."""
    from faceoff_amd.engine import VQVAEEngine
    img, gt = make_batch(7, 2, 4, 128, 128)
    img = torch.from_numpy(img).reshape(8, 6, 128, 128).cuda()
    gt = torch.from_numpy(gt).reshape(8, 3, 128, 128).cuda()
    res = []
    for direct in (False, True):
        if direct:
            os.environ["FACEOFF_NO_WINOGRAD"] = "1"
        try:
            eng = VQVAEEngine(make_state_dict(3, codebook_scale=0.3, gain=2.0), "cuda:0")
        finally:
            os.environ.pop("FACEOFF_NO_WINOGRAD", None)
        assert eng.winograd == (not direct)
        recon, diff, S = eng.loss_and_backward(img, gt, T=4)
        torch.cuda.synchronize()
        res.append((S["dec"].clone(), S["id_t"].clone(), S["id_b"].clone(), recon.item(), diff.item(),
                    {k: eng.flat_grads[o:o + n].clone() for k, (o, n) in eng.offsets.items()}))
    w, d = res
    assert torch.equal(w[1], d[1]) and torch.equal(w[2], d[2])
    assert (w[0] - d[0]).abs().max().item() <= 2e-5 * d[0].abs().max().item()
    np.testing.assert_allclose([w[3], w[4]], [d[3], d[4]], rtol=1e-5)
    for k in d[5]:
        err = (w[5][k] - d[5][k]).abs()
        scale = d[5][k].abs().max().item() + 1e-30
        assert err.max().item() <= 1e-3 * scale, (k, err.max().item() / scale)
    worst = sorted((w[5][k] - d[5][k]).abs().max().item() / (d[5][k].abs().max().item() + 1e-30) for k in d[5])
    assert worst[len(worst) // 2] <= 1e-4, worst[len(worst) // 2]        # the typical tensor is an order of magnitude closer


def test_three_training_steps_track_the_oracle():
    """Parity over consecutive iterations (zero_grad -> forward -> backward -> Adam, EMA codebooks moving; reference
    train_faceoff_perceptual.py:93-107): the losses of steps 2 and 3 depend on step 1's parameter update and codebook
    update.  Adam's first steps move every parameter by ~lr * sign(g), so a gradient that is zero up to rounding may move
    its parameter the other way (2 lr apart): losses and EMA codebooks within 1e-3 / 2e-3 at every step, parameters after three
    steps within 3 x 2.1 lr everywhere and within 3e-5 for at least 99 % of the 4 M elements."""
    from faceoff_amd.engine import VQVAEEngine
    from faceoff_amd.trainer import FaceOffTrainer
    from oracle import faceoff_oracle as O
    B, T, H, W = 2, 2, 64, 64
    sd = make_state_dict(6, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(21, B, T, H, W)
    p, st = O.to_torch_state(sd), {}
    eng = VQVAEEngine(sd, "cuda:0")
    tr = FaceOffTrainer(eng, lr=3e-4)
    x, y = torch.from_numpy(img).cuda(), torch.from_numpy(gt).cuda()
    for step in range(3):
        r = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), p, adam_state=st)
        recon, latent, _ = tr.step(x, y)
        np.testing.assert_allclose([recon.item(), latent.item()], [r["recon"].item(), r["latent"].item()], rtol=1e-3, err_msg=f"step {step}")
        for k in eng.buffers:
            assert _rel(eng.buffers[k].cpu().numpy(), p[k].numpy()) < 2e-3, (step, k)
    lr = 3e-4
    off = tot = 0
    for k, g in r["grads"].items():
        d = (eng.params[k].cpu() - p[k].detach()).abs()
        assert d.max().item() <= 3 * 2.1 * lr, k                      # at worst one sign flip per step
        off += int((d > 3e-5).sum())
        tot += d.numel()
    assert off <= 0.01 * tot, (off, tot)                              # ... and that is rare


def test_reference_goldens_with_the_resblock_halo_kernels_forced():
    """At the goldens' sizes (1-2 clips) the ResBlock family stays on the tiled kernels -- the halo-tile kernels (csrc/resblock_halo.hip,
    resblock_bwd.hip, resblock_bf16.hip) are taken from four tiles per CU up.  Here the same golden / oracle tests run once more in a fresh
    process with FACEOFF_FORCE_RESBLOCK_HALO=1 (the switch is read once per process), so that the kernels the timed configuration runs are
    held to the reference's outputs, code indices, losses and 70 gradients directly, not only through their agreement with the tiled forms."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FACEOFF_FORCE_RESBLOCK_HALO="1")
    sel = ("tests/test_e2e_gpu.py::test_c1_e2e_vs_reference_golden", "tests/test_e2e_gpu.py::test_c2_oneclip_vs_reference_golden",
           "tests/test_e2e_gpu.py::test_three_training_steps_track_the_oracle",
           "tests/test_bf16_engine_gpu.py::test_bf16_engine_step_vs_bf16_simulated_oracle_and_vs_fp32_oracle")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", *sel], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
