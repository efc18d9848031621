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
    want_gt = {"translate_horizontal": 0, "translate_vertical": 0, "rotate_image": 0}
    for name, v in P.draw_composite(r2):          # (the draw order itself is checked against the reference in test_oracle_vs_reference.py)
        want_gt[name if name.startswith("translate") else "rotate_image"] = v[0] if name == "distort_image" else v
    assert out.shape == xg.shape and gt == want_gt and r1.random() == r2.random()


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


def _u8(shape, seed):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def test_u8_perturbations_equal_the_opencv_restatement_bit_for_bit():
    """uint8 frames [N,H,W,C] through csrc/warp_u8.hip against oracle/cv2_oracle.py (OpenCV 4.6.0's 8-bit warpAffine / resize /
    flip restated): every byte equal.  Covers the reference's parameter ranges (perturbations.py:271-295 and ranges.py), fractional and
    out-of-image translations, shear, every magnification 0.90 ... 1.10 and a few outside, 1 / 3 / 4 channels, odd sizes."""
    from faceoff_amd import perturbations as P
    from oracle import cv2_oracle as O
    for shape, seed in (((2, 40, 56, 3), 0), ((1, 33, 47, 4), 1), ((2, 64, 48, 1), 2)):
        x = _u8(shape, seed)
        xg = torch.from_numpy(x).cuda()
        N, H, W, _ = shape

        def same(got, fn):
            got = got.cpu().numpy()
            for n in range(N):
                want = fn(x[n] if shape[3] > 1 else x[n, ..., 0])
                assert np.array_equal(got[n] if shape[3] > 1 else got[n, ..., 0], want), (shape, n)

        for t in (-20, -3, 0, 1, 3, 20, 0.5, -7.3, W + 5, -(W + 5)):
            same(P.translate_horizontal(t, xg), lambda im: O.translate_horizontal(t, im))
            same(P.translate_vertical(t, xg), lambda im: O.translate_vertical(t, im))
        for angle in (-25, -3, -1, 0, 1, 2, 3, 7, 25, 45, 90, 180, 33.3):
            same(P.rotate_image(angle, xg), lambda im: O.rotate_image(angle, im))
            same(P.rotate_image(angle, xg, center=(20, 17)), lambda im: O.rotate_image(angle, im, center=(20, 17)))
            same(P.rotate_image(angle, xg, center=(20.5, 17.25)), lambda im: O.rotate_image(angle, im, center=(20.5, 17.25)))
        for sh in (-0.1, 0.05, 0.1):
            same(P.shear_image(sh, xg), lambda im: O.shear_image(sh, im))
        for k in list(range(90, 111)) + [50, 75, 150, 200]:
            same(P.resize_image(k / 100, xg), lambda im: O.resize_image(k / 100, im))
        for code in (0, 1, -1):
            same(P.image_flip(code, xg), lambda im: O.image_flip(code, im))
    # a single [H,W,C] image; numpy scalars as parameters and centre (find_eye_center returns numpy ints, perturbations.py:183-196)
    x = _u8((40, 56, 3), 9)
    assert np.array_equal(P.rotate_image(5, torch.from_numpy(x).cuda()).cpu().numpy(), O.rotate_image(5, x))
    c = np.array([21.6, 17.2]).astype("int")
    got = P.rotate_image(np.float64(-3.0), torch.from_numpy(x).cuda(), center=c).cpu().numpy()
    assert np.array_equal(got, O.rotate_image(-3.0, x, center=(21, 17)))
    assert np.array_equal(P.translate_horizontal(np.int64(2), torch.from_numpy(x).cuda()).cpu().numpy(), O.translate_horizontal(2, x))


def test_u8_perturbations_per_frame_parameters_at_full_size():
    """a loader batch of 160 frames 256x256x3 (BASELINE configs[1]: bs 32 x T 5), one parameter per frame as the reference draws
    them (TemporalAlignment/dataset.py:34-54): more frames than one launch carries, checked on sampled frames."""
    from faceoff_amd import perturbations as P
    from oracle import cv2_oracle as O
    N = 160
    x = _u8((N, 256, 256, 3), 11)
    xg = torch.from_numpy(x).cuda()
    r = random.Random(5)
    tx = [r.randint(-20, 20) for _ in range(N)]
    rot = [r.randint(-25, 25) for _ in range(N)]
    mag = [r.randint(90, 110) / 100 for _ in range(N)]
    centers = [(r.randint(100, 150), r.randint(90, 140)) for _ in range(N)]
    a = P.translate_horizontal(tx, xg).cpu().numpy()
    b = P.rotate_image(rot, xg, center=centers).cpu().numpy()
    c = P.resize_image(mag, xg).cpu().numpy()
    for n in (0, 15, 16, 63, 64, 65, 127, 128, 159):
        assert np.array_equal(a[n], O.translate_horizontal(tx[n], x[n]))
        assert np.array_equal(b[n], O.rotate_image(rot[n], x[n], center=centers[n]))
        assert np.array_equal(c[n], O.resize_image(mag[n], x[n]))
    with pytest.raises(ValueError):
        P.translate_horizontal([1, 2, 3], xg)
    # the composite on uint8 frames: same draws as on float frames, applied with the OpenCV arithmetic
    r1, r2 = random.Random(3), random.Random(3)
    out, gt = P.perturb_image_composite(xg[:5], (128, 110), rng=r1)
    want = x[:5]
    fns = {"translate_horizontal": O.translate_horizontal, "translate_vertical": O.translate_vertical, "rotate_image": O.rotate_image,
           "resize_image": O.resize_image}
    for name, v in P.draw_composite(r2):
        if name != "distort_image":               # Wand: not built, the frames pass through
            want = np.stack([fns[name](v, f, center=(128, 110)) if name == "rotate_image" else fns[name](v, f) for f in want])
    assert np.array_equal(out.cpu().numpy(), want)


def test_u8_frames_to_normalised_tensors():
    """ToTensor + Normalize (TemporalAlignment/dataset.py:244-256): float(v) / 255, then (t - 0.5) / 0.5, bit for bit."""
    from faceoff_amd import perturbations as P
    x = _u8((3, 24, 40, 3), 12)
    xg = torch.from_numpy(x).cuda()
    t = torch.from_numpy(x).permute(0, 3, 1, 2).to(torch.float32).div(255)
    want = t.sub(torch.tensor(0.5)).div(torch.tensor(0.5))
    assert torch.equal(P.to_normalized(xg).cpu(), want)
    assert torch.equal(P.to_normalized(xg, reverse_channels=True).cpu(), want.flip(1))
    m, s = 0.485, 0.229
    want = t.sub(torch.tensor(m)).div(torch.tensor(s))
    assert torch.equal(P.to_normalized(xg, mean=m, std=s).cpu(), want)


def test_filter_gradient_stream_is_kept_off_the_main_streams_hardware_queue(monkeypatch):
    """HIP hands out hardware queues in order of first use (2, 3, 4, 4, 3, 2, 1, ..; 1 is the null stream's), so with other streams used first -- another
    engine's, a communicator's -- the engine's filter-gradient stream can land on the main stream's queue, where its launches line up behind the chain they are
    meant to run beside (config 2 +3.6 %).  Emulated with three throw-away streams in front: the trainer's check replaces the stream until a launch on it
    finishes while long launches occupy the main stream."""
    from faceoff_amd.engine import VQVAEEngine, _runs_beside_current
    from faceoff_amd.synth import make_state_dict
    from faceoff_amd.trainer import FaceOffTrainer
    dev = torch.device("cuda:0")
    monkeypatch.setenv("FACEOFF_DIAG_QUEUE_SHIFT", "3")
    monkeypatch.setenv("FACEOFF_NO_QUEUE_CHECK", "1")
    eng = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
    FaceOffTrainer(eng, lr=3e-4)
    unchecked = _runs_beside_current(eng.wgrad_stream, dev)          # (whatever the process history made of it: recorded, not asserted)
    monkeypatch.delenv("FACEOFF_NO_QUEUE_CHECK")
    eng2 = VQVAEEngine(make_state_dict(0, codebook_scale=0.3, gain=2.0), dev)
    FaceOffTrainer(eng2, lr=3e-4)
    assert _runs_beside_current(eng2.wgrad_stream, dev), f"the checked stream still shares the main stream's queue (unchecked engine: beside = {unchecked})"
    assert eng2._streams[0] is eng2.wgrad_stream
