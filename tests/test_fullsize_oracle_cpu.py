"""tests/_fullsize_oracle.py (the CPU oracle evaluated clip chunk by clip chunk, used by the -m gpu tests at the timed sizes)
against oracle.train_step on the whole batch, at a size where both run here: same codes, same losses, gradients and EMA buffers to
fp32 summation order -- with and without the LPIPS branch, the bf16 rounding points and forced codes."""
import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch, make_vgg_lpips_state
from oracle import faceoff_oracle as O
from _fullsize_oracle import oracle_step_chunked


@pytest.mark.parametrize("mode", ["c2", "c3", "c3_forced"])
def test_chunked_oracle_equals_the_oracle_on_the_whole_batch(mode):
    B, T, H, W = 5, 2, 32, 32                                  # chunks of 2, 2, 1 clips
    sd = make_state_dict(3, codebook_scale=0.3, gain=2.0)
    img, gt = (torch.from_numpy(a) for a in make_batch(77, B, T, H, W))
    kw = {}
    lp = None
    if mode != "c2":
        lp = make_vgg_lpips_state(7)
        kw = dict(bf16sim=True, lpips_bf16sim=True)
    p = O.to_torch_state(sd)
    whole = O.train_step(img, gt, p, lpips_state=None if lp is None else {k: torch.from_numpy(v) for k, v in lp.items()}, **kw)
    force = None
    if mode == "c3_forced":                                    # any fixed codes: the oracle's own, rolled by one
        force = (torch.roll(whole["fw"]["id_t"], 1, 0), torch.roll(whole["fw"]["id_b"], 1, 0))
        p = O.to_torch_state(sd)
        whole = O.train_step(img, gt, p, lpips_state={k: torch.from_numpy(v) for k, v in lp.items()}, force_ids=force, **kw)
    got = oracle_step_chunked(img, gt, sd, lpips_state=lp, force_ids=force, clips_per_chunk=2, threads=4, **kw)
    assert torch.equal(got["id_t"], whole["fw"]["id_t"]) and torch.equal(got["id_b"], whole["fw"]["id_b"])
    wdec = whole["fw"]["dec"].detach()
    # fp32: oneDNN blocks a 4-frame and a 10-frame batch differently (not the same bits, 1e-6).  With the bf16 rounding points that
    # summation noise flips isolated bf16 roundings (1 ulp = 0.4 %) and through them ReLU masks: the bf16-simulated oracle agrees with
    # ITSELF under re-batching only at bf16 level -- the floor under every bf16 comparison in tests/test_c3_gpu.py, printed here.
    bf16 = mode != "c2"
    rl2 = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rl2(got["dec"], wdec) <= (2e-3 if bf16 else 1e-6)
    for k in ("recon", "latent", "perceptual"):
        np.testing.assert_allclose(got[k], float(whole[k].detach()), rtol=2e-3 if bf16 else 2e-6, atol=1e-9)
    errs = sorted(((rl2(got["grads"][n], g), n) for n, g in whole["grads"].items()), reverse=True)
    print(f"[chunked vs whole oracle, {mode}] dec rel-L2 {rl2(got['dec'], wdec):.2e}; gradients worst {errs[0]}, median {errs[len(errs) // 2][0]:.2e}")
    assert errs[0][0] <= (8e-2 if bf16 else 2e-5), errs[0]
    assert errs[len(errs) // 2][0] <= (2e-2 if bf16 else 5e-6)
    for n, b in got["buffers"].items():                        # `p` holds the whole-batch step's post-EMA buffers
        assert rl2(b, p[n]) <= (2e-3 if bf16 else 1e-5), n
