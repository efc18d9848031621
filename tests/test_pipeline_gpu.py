"""Device-side input pipeline and validation path: the affine perturbations against the numpy checker
(oracle/warp_oracle.py), properties cv2 shares (integer translations are exact shifts with zero fill, rotation by 0 and scale 1
are the identity), the validation loop + 8-bit de-normalisation, the asynchronous metric accumulator."""
import math
import random

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict

pytestmark = pytest.mark.gpu


def test_affine_perturbations_vs_numpy_checker():
    from faceoff_amd import perturbations as P
    from oracle import warp_oracle as WO
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, (3, 3, 40, 56)).astype(np.float32)
    xg = torch.from_numpy(x).cuda()
    # integer translations: exact shifts, zeros shifted in
    y = P.translate_horizontal(3, xg).cpu().numpy()
    assert np.array_equal(y[..., 3:], x[..., :-3]) and np.all(y[..., :3] == 0)
    y = P.translate_vertical(-2, xg).cpu().numpy()
    assert np.array_equal(y[..., :-2, :], x[..., 2:, :]) and np.all(y[..., -2:, :] == 0)
    assert torch.equal(P.rotate_image(0, xg), xg)
    np.testing.assert_allclose(P.resize_image(1.0, xg).cpu().numpy(), x, atol=1e-6)
    # rotation about the centre and about an eye centre; zoom in / out (bicubic)
    for angle, center in ((7.0, None), (-3.0, (20.5, 17.25))):
        got = P.rotate_image(angle, xg, center=center).cpu().numpy()
        c = (56 // 2, 40 // 2) if center is None else center
        want = WO.warp_affine(x, P.rotation_matrix(c, angle))
        assert np.abs(got - want).max() <= 2e-4
    for m in (1.1, 0.9, 1.07):
        got = P.resize_image(m, xg).cpu().numpy()
        ws, hs = int(round(56 * m)), int(round(40 * m))
        if m >= 1:
            fwd = (ws / 56, 0, (0.5 * ws / 56 - 0.5) - (ws // 2 - 28), 0, hs / 40, (0.5 * hs / 40 - 0.5) - (hs // 2 - 20))
            want = WO.warp_affine(x, fwd, mode=1)
        else:
            ox, oy = (56 - ws) // 2, (40 - hs) // 2
            want = WO.warp_affine(x, (ws / 56, 0, (0.5 * ws / 56 - 0.5) + ox, 0, hs / 40, (0.5 * hs / 40 - 0.5) + oy), mode=1)
            mask = np.zeros((40, 56)); mask[oy:oy + hs, ox:ox + ws] = 1
            want = want * mask
        assert np.abs(got - want).max() <= 5e-4, m
    # the composite draws its parameters like the reference (:236-262): same random stream -> same choices
    r1, r2 = random.Random(3), random.Random(3)
    out, gt = P.perturb_image_composite(xg, (28.0, 18.0), rng=r1)
    chosen = []
    while not chosen:
        chosen = [i for i in range(4) if r2.randint(0, 1)]
    want_gt = {"translate_horizontal": 0, "translate_vertical": 0, "rotate_image": 0}
    ranges = [(-3, 3, 1), (-3, 3, 1), (-3, 3, 1), (90, 110, 100)]                 # TemporalAlignment/ranges.py
    for i in chosen:
        v = r2.randint(ranges[i][0], ranges[i][1]) / ranges[i][2]
        want_gt[("translate_horizontal", "translate_vertical", "rotate_image", "rotate_image")[i]] = v
    assert out.shape == xg.shape and gt == want_gt


def test_validation_loop_and_denormalisation():
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    from faceoff_amd.validation import validation, denormalize_u8
    g = torch.Generator().manual_seed(1)
    loader = [tuple(torch.rand((1, 3, 3, 32, 32), generator=g) * 3 - 1.5 for _ in range(5)) for _ in range(2)]
    model = VQVAE(in_channel=6).to("cuda")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(1, codebook_scale=0.3, gain=2.0).items()})
    model.train()
    emb = model.state_dict()["quantize_b.embed"].clone()
    res = validation(model, loader, "cuda")
    assert model.training and len(res) == 2
    assert torch.equal(model.state_dict()["quantize_b.embed"], emb)              # eval mode: no EMA update (:59)
    src, tgt, bg, si, so = loader[0]
    want = lambda t: ((t.clamp(-1, 1) + 1) / 2 * 255).permute(0, 2, 3, 1).numpy().astype(np.uint8)
    assert np.array_equal(res[0]["source"].cpu().numpy(), want(src[0]))
    assert np.array_equal(res[0]["background"].cpu().numpy(), want(bg[0]))
    assert np.array_equal(res[0]["source_images"].cpu().numpy(), want(si[0]))
    model.eval()
    with torch.no_grad():
        out, _ = model(torch.cat([src, bg], dim=2).squeeze(0).cuda())
    assert np.abs(res[0]["prediction"].cpu().numpy().astype(int) - want(out[:, :3].cpu()).astype(int)).max() <= 1
    nh = torch.zeros((3, 32, 32, 8), device="cuda")
    nh[..., :3] = src[0].permute(0, 2, 3, 1).cuda()
    assert np.array_equal(denormalize_u8(nh, channels_last_ld=8, bgr=True).cpu().numpy(), want(src[0])[..., ::-1])


def test_metric_accumulator_matches_the_trainer_average():
    from faceoff_amd.validation import MetricAccumulator
    m = MetricAccumulator("cuda")
    vals = [(0.5, 5), (0.25, 3), (1.0, 30)]
    for v, s in vals:
        m.update(torch.tensor([v], device="cuda"), s)
    assert math.isclose(m.average(), sum(v * s for v, s in vals) / sum(s for _, s in vals), rel_tol=1e-6)
