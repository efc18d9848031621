"""BASELINE config 3 as SURVEY.md section 8(d) defines it -- "bf16 MFMA inputs / fp32 accumulate & master weights" for the VQ-VAE itself --
on a real MI355X: the engine with dtype="bf16" against

  (1) the oracle with the SAME rounding points (oracle.faceoff_oracle._BF16Sim: every conv input, filter and stored activation / activation
      gradient rounded to bfloat16 once; accumulation, biases, VQ distances / arg-min / commitment loss, losses and filter gradients fp32).
      What is left between the two is fp32 summation order, which flips isolated bf16 roundings (1 ulp = 0.4 %) and, through them, ReLU masks
      and VQ near-ties: code indices are margin-gated against THAT oracle, gradients compared by relative L2;
  (2) the pure-fp32 oracle (the reference's arithmetic): reported, and bounded by what the bf16-simulated oracle itself deviates from it
      ("1e-3 vs an fp32 oracle is not attainable with bf16 operands ... state both").

Flip-free evidence: the TEACHER-FORCED step -- engine and oracle both use the oracle's code indices, so no VQ near-tie can open
an O(1) gap -- holds the decoder output to 1e-2 and every one of the 70 gradient tensors to twice its recorded error (tests/_observed.py); the
free-running step compares the decoder output OUTSIDE the receptive fields of its (margin-gated) flipped codes at the same 1e-2.

Sizes: 2 clips x 2 frames of 64x64 (BASELINE config 1's shape) and a ragged 3 x 3 x 40x24 (gather-form filter gradients, tile tails);
at the timed size (160 frames of 256x256) size-independent properties: finite, bit-reproducible, every code a true nearest code of its fp32
input, and agreement with the fp32 engine at bf16 level."""
import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch
from _observed import Observed

pytestmark = pytest.mark.gpu
FIXTURES = [(2, 2, 64, 64, 0), (3, 3, 40, 24, 11)]


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _step(sd, img, gt, B, T, H, W, dtype="bf16", force_ids=None):
    from faceoff_amd.engine import VQVAEEngine
    eng = VQVAEEngine(sd, "cuda:0", dtype=dtype)
    x = torch.from_numpy(img).reshape(B * T, 6, H, W).cuda()
    y = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
    recon, diff, S = eng.loss_and_backward(x, y, T=T, force_ids=force_ids)
    torch.cuda.synchronize()
    return eng, recon, diff, S


def _dilate(m, r):
    """m: bool [N,h,w]; a (2r+1)^2 box dilation"""
    return torch.nn.functional.max_pool2d(m.float().unsqueeze(1), 2 * r + 1, 1, r).squeeze(1) > 0


def _outside_flipped_receptive_fields(bad_t, bad_b):
    """bool [N,H,W] at image resolution: the pixels of `dec` that no flipped code can reach.  A top code (1/8 resolution) enters the decoder
    through upsample_t (k4 s2 p1: latent rows 2y-1 .. 2y+2); a bottom code (1/4 resolution) directly.  From there `dec` (:218-225) is a 3x3
    conv, two ResBlocks (3x3 each) and two k4 s2 p1 transposed convs: 3 latent pixels + 1 per transposed stage -- 5 latent pixels on each
    side are excluded (one to spare)."""
    up = _dilate(bad_t, 1).repeat_interleave(2, 1).repeat_interleave(2, 2)
    reach = _dilate(bad_b | up, 5)
    return ~reach.repeat_interleave(4, 1).repeat_interleave(4, 2)


@pytest.mark.parametrize("B,T,H,W,seed", FIXTURES)
def test_bf16_engine_teacher_forced_step_vs_bf16_simulated_oracle(B, T, H, W, seed):
    """The flip-free comparison: the engine runs the step on the bf16-simulated oracle's own code indices (forward(force_ids=...): the search
    is skipped, gather / straight-through / commitment loss / EMA statistics use the given codes).  What separates the two sides is then only
    fp32 summation order and the isolated bf16 roundings / ReLU masks it flips -- no code can differ, so nothing opens an O(1) gap:
    decoder output <= 1e-2, losses 2e-3, EMA buffers 5e-3, and EVERY gradient tensor within twice its recorded error."""
    from faceoff_amd import ops
    from oracle import faceoff_oracle as O
    sd = make_state_dict(seed, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(1234 + seed, B, T, H, W)
    r = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), O.to_torch_state(sd), bf16sim=True)
    ids = (r["fw"]["id_t"], r["fw"]["id_b"])
    # (the oracle forced onto its own codes is the oracle: checked once here, so that `r` IS the teacher-forced reference)
    r2 = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), O.to_torch_state(sd), bf16sim=True, force_ids=ids)
    assert torch.equal(r2["fw"]["dec"], r["fw"]["dec"]) and all(torch.equal(r2["grads"][k], g) for k, g in r["grads"].items())
    eng, recon, diff, S = _step(sd, img, gt, B, T, H, W, force_ids=ids)
    assert torch.equal(S["id_t"].cpu(), ids[0]) and torch.equal(S["id_b"].cpu(), ids[1])
    obs = Observed(f"forced_{B}x{T}x{H}x{W}")
    np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=2e-3)
    np.testing.assert_allclose(diff.item(), r["latent"].item(), rtol=2e-3)
    dec_err = _rel_l2(ops.nhwc_to_nchw(S["dec"], 6).cpu(), r["fw"]["dec"].detach())
    obs.check("dec", dec_err, cap=1e-2)
    errs = sorted(((_rel_l2(eng.grads[k].cpu(), g), k) for k, g in r["grads"].items()), reverse=True)
    print(f"[bf16 engine, teacher-forced {B}x{T}x{H}x{W}] dec rel L2 {dec_err:.2e}; gradients vs bf16-simulated oracle: worst {errs[0][0]:.2e} ({errs[0][1]}), "
          f"median {errs[len(errs) // 2][0]:.2e}, best {errs[-1][0]:.2e} ({errs[-1][1]})")
    for e, k in errs:
        obs.check("grad:" + k, e, cap=6e-2)
    for k in eng.buffers:
        np.testing.assert_allclose(eng.buffers[k].cpu().numpy(), r["fw"]["new_buffers"][k].numpy(), rtol=5e-3, atol=5e-3 * float(r["fw"]["new_buffers"][k].abs().max()))
    obs.flush()


@pytest.mark.parametrize("B,T,H,W,seed", FIXTURES)
def test_bf16_engine_step_vs_bf16_simulated_oracle_and_vs_fp32_oracle(B, T, H, W, seed):
    from faceoff_amd import ops
    from oracle import faceoff_oracle as O
    sd = make_state_dict(seed, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(1234 + seed, B, T, H, W)
    ref = {}
    for name, sim in (("bf16sim", True), ("fp32", False)):
        ref[name] = O.train_step(torch.from_numpy(img), torch.from_numpy(gt), O.to_torch_state(sd), bf16sim=sim)
    eng, recon, diff, S = _step(sd, img, gt, B, T, H, W)
    assert S["dec"].dtype == torch.float32 and S["eb"].dtype == torch.bfloat16 and S["qb_in"].dtype == torch.float32
    dec = ops.nhwc_to_nchw(S["dec"], 6).cpu()
    r = ref["bf16sim"]
    # losses: fp32 reductions of values that agree to bf16 rounding noise
    np.testing.assert_allclose(recon.item(), r["recon"].item(), rtol=2e-3)
    np.testing.assert_allclose(diff.item(), r["latent"].item(), rtol=2e-3)
    # code indices: equal, or a near-tie of the bf16-simulated oracle's own distances (its top-2 margin below 1e-2: the quantiser's input
    # carries ~1e-3 of summation-order noise per element after a dozen bf16-rounded layers; typical margins are ~0.2)
    flips, badmap = 0, {}
    for lvl in "tb":
        bad = (S["id_" + lvl].cpu() != r["fw"]["id_" + lvl]).reshape(-1)
        badmap[lvl] = bad.reshape(r["fw"]["id_" + lvl].shape).clone()
        if lvl == "b" and flips:
            # a flipped TOP code (a near-tie, gated in the previous round of this loop) is decoded into the bottom quantiser's input: around
            # it that input differs by O(1) and the bottom codes with it.  Those positions (at most a 24 x 24 neighbourhood per flipped
            # top code: dec_t's three 3x3 stages and its stride-2 stem) are not compared; every other mismatch is gated on the margin.
            ref_in = r["fw"]["qb_in"].detach()
            moved = (S["qb_in"].cpu() - ref_in).abs().reshape(-1, 64).max(1).values > 3e-2 * ref_in.abs().max()
            assert int(moved.sum()) <= 24 * 24 * flips, (int(moved.sum()), flips)
            bad = bad & ~moved
        margin = O.vq_margin(r["fw"][f"q{lvl}_in"].detach(), torch.from_numpy(sd[f"quantize_{lvl}.embed"]))
        assert bool((margin[bad] < 1e-2).all()) and bad.float().mean().item() < 1e-2, (lvl, int(bad.sum()), margin[bad].max().item() if bad.any() else 0)
        flips += int(bad.sum())
    obs = Observed(f"free_{B}x{T}x{H}x{W}")
    # decoder output: a flipped code changes a patch by O(1) -- compared OUTSIDE the flipped codes' receptive fields, at the flip-free bound
    keep = _outside_flipped_receptive_fields(badmap["t"], badmap["b"]).unsqueeze(1).expand(-1, 3, -1, -1)
    assert keep.float().mean().item() > 0.3, "the flipped codes' receptive fields cover most of the fixture: nothing left to compare"
    ref_dec = r["fw"]["dec"].detach()[:, :3]
    dec_err = _rel_l2(dec[:, :3][keep], ref_dec[keep])
    dec_all = _rel_l2(dec, r["fw"]["dec"].detach())
    obs.check("dec_outside_flips", dec_err, cap=1e-2)
    # all 70 gradients, relative L2 per tensor: filter gradients are sums over every position, the flipped codes' patches included, so the
    # per-tensor statement lives in the teacher-forced test above; here the aggregate is held to twice what this fixture was recorded at
    errs = sorted(((_rel_l2(eng.grads[k].cpu(), g), k) for k, g in r["grads"].items()), reverse=True)
    med = errs[len(errs) // 2][0]
    errs32 = sorted(((_rel_l2(eng.grads[k].cpu(), g), k) for k, g in ref["fp32"]["grads"].items()), reverse=True)
    sim32 = sorted(((_rel_l2(r["grads"][k], g), k) for k, g in ref["fp32"]["grads"].items()), reverse=True)
    print(f"[bf16 engine {B}x{T}x{H}x{W}] index flips {flips}; dec rel L2 {dec_all:.2e} ({dec_err:.2e} outside the flipped codes' receptive fields, "
          f"{keep.float().mean().item():.2f} of the pixels); gradients vs bf16-simulated oracle: worst {errs[0][0]:.2e} ({errs[0][1]}), "
          f"median {med:.2e}; vs fp32 oracle: worst {errs32[0][0]:.2e}, median {errs32[len(errs32) // 2][0]:.2e}; "
          f"bf16-simulated oracle vs fp32 oracle: worst {sim32[0][0]:.2e}, median {sim32[len(sim32) // 2][0]:.2e}")
    obs.check("grad_median", med)
    obs.check("grad_worst", errs[0][0])
    obs.flush()
    # the rounding points are the oracle's: the engine is no further from the fp32 arithmetic than the bf16-simulated oracle is (within 25 %)
    assert errs32[len(errs32) // 2][0] <= (1.25 if flips == 0 else 2.0) * sim32[len(sim32) // 2][0] + 1e-3
    # EMA codebook buffers (fp32 statistics of fp32 inputs; a flipped code moves two rows)
    for k in eng.buffers:
        np.testing.assert_allclose(eng.buffers[k].cpu().double().norm().item(), r["fw"]["new_buffers"][k].double().norm().item(),
                                   rtol=5e-3 if flips == 0 else 5e-2)


def test_bf16_engine_kernel_paths_agree(monkeypatch):
    """The same step with the 256-row ping-pong kernels forced at this small size (FACEOFF_BF16_BIG_TILES) and with the 128-row kernels:
    same bf16 operands, different tile shapes and summation orders -> identical code indices, activations within a bf16 ulp, filter gradients
    to 2 % (isolated rounding flips of stored activations)."""
    B, T, H, W = 2, 3, 64, 64
    sd = make_state_dict(5, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(77, B, T, H, W)
    e0, r0, d0, S0 = _step(sd, img, gt, B, T, H, W)
    monkeypatch.setenv("FACEOFF_BF16_BIG_TILES", "1")
    e1, r1, d1, S1 = _step(sd, img, gt, B, T, H, W)
    np.testing.assert_allclose([r0.item(), d0.item()], [r1.item(), d1.item()], rtol=2e-3)
    same = [(S0["id_" + l] == S1["id_" + l]).float().mean().item() for l in "tb"]
    assert min(same) > 0.99, same
    # activations: in front of the quantisers always; behind them only where every code agrees (the two kernel families walk K in different
    # orders -- (tap, chunk) against (chunk, taps) -- so a near-tie of the quantiser's input may resolve the other way: `same` bounds how often)
    for k in ("a1", "eb", "c1", "d3") + (("u2", "v2") if min(same) == 1.0 else ()):
        a, b = S0[k].float(), S1[k].float()
        assert _rel_l2(a, b) < 1e-2, k
    worst = max((_rel_l2(e0.grads[k], e1.grads[k]), k) for k in e0.grads)
    print(f"[bf16 engine, 128-row vs 256-row kernels] code agreement {same}, worst gradient rel L2 {worst}")
    assert worst[0] < (2e-2 if min(same) == 1.0 else 0.2), worst


def test_bf16_engine_at_the_timed_size_properties():
    """160 frames of 256x256, T = 5 (BASELINE config 2 / 3 size): finite; gradients bit-reproducible run to run (fixed-order slab sums, no
    atomics in any filter-gradient path); every chosen code is a true nearest code of the quantiser's fp32 input (fp64 check); and the step
    agrees with the fp32 engine on the same inputs at bf16 level (losses 1 %; > 90 % of the 819 200 codes -- a top-level near-tie that
    resolves the other way re-decodes up to 24 x 24 bottom positions around it, so bottom-level agreement is several times the raw flip
    rate away from 1; gradient direction)."""
    from faceoff_amd.engine import VQVAEEngine
    from oracle import faceoff_oracle as O
    B, T, H = 32, 5, 256
    sd = make_state_dict(0, codebook_scale=0.3, gain=2.0)
    gen = torch.Generator(device="cuda").manual_seed(99)
    x = torch.rand((B * T, 6, H, H), device="cuda", generator=gen) * 2 - 1
    y = torch.rand((B * T, 3, H, H), device="cuda", generator=gen) * 2 - 1
    runs = []
    for rep in range(2):
        eng = VQVAEEngine(sd, "cuda:0", dtype="bf16")
        recon, diff, S = eng.loss_and_backward(x, y, T=T)
        torch.cuda.synchronize()
        runs.append((recon.item(), diff.item(), eng.flat_grads.clone(), S["id_t"].clone(), S["id_b"].clone()))
        if rep == 0:
            assert torch.isfinite(eng.flat_grads).all() and np.isfinite(recon.item()) and np.isfinite(diff.item())
            for lvl, qin in (("t", S["qt_in"]), ("b", S["qb_in"])):        # nearest-code property on a 65 536-vector sample
                v = qin.reshape(-1, 64)[:: max(1, qin.numel() // 64 // 65536)].double()
                ids = S["id_" + lvl].reshape(-1)[:: max(1, qin.numel() // 64 // 65536)]
                e = torch.from_numpy(sd[f"quantize_{lvl}.embed"]).cuda().double()
                dist = v.pow(2).sum(1, keepdim=True) - 2 * v @ e + e.pow(2).sum(0, keepdim=True)
                chosen = dist.gather(1, ids.reshape(-1, 1)).squeeze(1)
                assert bool((chosen <= dist.min(1).values + 1e-5).all()), lvl
        del eng, S
    assert torch.equal(runs[0][2], runs[1][2]) and torch.equal(runs[0][3], runs[1][3]) and torch.equal(runs[0][4], runs[1][4])
    e32 = VQVAEEngine(sd, "cuda:0", dtype="fp32")
    r32, d32, S32 = e32.loss_and_backward(x, y, T=T)
    torch.cuda.synchronize()
    np.testing.assert_allclose([runs[0][0], runs[0][1]], [r32.item(), d32.item()], rtol=1e-2)
    agree = [(runs[0][3] == S32["id_t"]).float().mean().item(), (runs[0][4] == S32["id_b"]).float().mean().item()]
    cos = torch.nn.functional.cosine_similarity(runs[0][2].double(), e32.flat_grads.double(), dim=0).item()
    print(f"[bf16 vs fp32 engine at 160 x 256x256] losses {runs[0][:2]} vs {(r32.item(), d32.item())}; code agreement {agree}; gradient cosine {cos:.5f}")
    assert agree[0] > 0.98 and agree[1] > 0.90 and cos > 0.98, (agree, cos)


def test_module_mirror_trains_and_infers_with_the_bf16_engine(monkeypatch):
    """FACEOFF_DTYPE=bf16: the drop-in nn.Module (reference API, fp32 NCHW tensors in and out, fp32 nn.Parameters) on the bf16-operand
    engine: a reference-style step (torch MSE loss, loss.backward(), torch.optim.Adam) equals the engine's fused step on the same data,
    and the staged inference entry points (only_encode / encode_quantized / decode / decode_code) reproduce the eval forward."""
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    monkeypatch.setenv("FACEOFF_DTYPE", "bf16")
    B, T, H, W = 2, 2, 64, 64
    sd = make_state_dict(3, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(41, B, T, H, W)
    model = VQVAE(in_channel=6).to("cuda")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    x = torch.from_numpy(img).cuda()
    y = torch.from_numpy(gt).reshape(B * T, 3, H, W).cuda()
    out, latent = model(x)
    assert model._engine.bf16 and out.dtype == torch.float32 and out.shape == (B * T, 6, H, W)
    loss = torch.nn.functional.mse_loss(out[:, :3], y) + latent.mean()
    loss.backward()
    eng, recon, diff, S = _step(sd, img, gt, B, T, H, W)
    np.testing.assert_allclose(loss.item(), recon.item() + diff.item(), rtol=1e-5)
    params = dict(model.named_parameters())
    worst = max(_rel_l2(params[k].grad, eng.grads[k]) for k in eng.grads)
    assert worst < 1e-5, worst             # (same kernels, same inputs: only the fp32 loss gradient takes a different route into g_dec)
    model.eval()
    with torch.no_grad():
        out2, _ = model(x)
        id_t, id_b = model._last_ids
        dec3 = model.decode_code(id_t, id_b)
        enc_b, enc_t = model.only_encode(x.reshape(B * T, 6, H, W))
    assert _rel_l2(dec3, out2) < 2e-2      # codes -> decode: the decoder input is the bf16-rounded codebook row in both
    assert enc_b.shape == (B * T, 128, H // 4, W // 4) and enc_b.min().item() >= 0.0 and enc_t.shape == (B * T, 128, H // 8, W // 8)
