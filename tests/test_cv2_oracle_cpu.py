"""The OpenCV restatement (oracle/cv2_oracle.py) against what its algorithm implies -- cv2 is not installed here and the reference
holds no fixtures for its perturbations, so these properties are what the checker of tests/test_pipeline_gpu.py is pinned on:
exact shifts for whole-pixel translations with zeros shifted in (BORDER_CONSTANT), (a + b + 1) >> 1 for a half-pixel translation
(1/32-pixel coordinates, 15-bit weights, round half up), identities at rotation 0 / zoom 1.0, a half turn about the centre pixel
equals cv2.flip(-1), the weight tables, the resize geometry of perturbations.py:87-105."""
import numpy as np

from oracle import cv2_oracle as O


def _img(shape, seed=0):
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def test_bilinear_table_is_opencvs():
    t = O.bilinear_tab()
    assert t.shape == (1024, 4) and (t.sum(1) == 32768).all()
    assert t[0].tolist() == [32767, 0, 0, 1]                   # saturate_cast<short>(32768), then the sum fix-up on the last weight
    assert t[1].tolist() == [31744, 1024, 0, 0] and t[33].tolist() == [30752, 992, 992, 32]
    assert t[16 * 32 + 16].tolist() == [8192] * 4


def test_whole_pixel_translations_are_shifts_with_zero_fill():
    img = _img((40, 56, 3))
    y = O.translate_horizontal(3, img)
    assert np.array_equal(y[:, 3:], img[:, :-3]) and not y[:, :3].any()
    y = O.translate_horizontal(-5, img)
    assert np.array_equal(y[:, :-5], img[:, 5:]) and not y[:, -5:].any()
    y = O.translate_vertical(-2, img)
    assert np.array_equal(y[:-2], img[2:]) and not y[-2:].any()
    assert not O.translate_horizontal(56, img).any()


def test_half_pixel_translation_rounds_half_up():
    img = _img((9, 17, 3), 1)
    y = O.warp_affine(img, np.float32([[1, 0, 0.5], [0, 1, 0]]))
    a, b = img[:, :-1].astype(int), img[:, 1:].astype(int)
    assert np.array_equal(y[:, 1:], ((a + b + 1) >> 1).astype(np.uint8))
    assert np.array_equal(y[:, 0], ((img[:, 0].astype(int) + 1) >> 1).astype(np.uint8))        # the left neighbour is the border: 0
    y = O.warp_affine(img, np.float32([[1, 0, 0], [0, 1, 0.25]]))
    a, b = img[:-1].astype(int), img[1:].astype(int)
    assert np.array_equal(y[1:], ((a * 8192 + b * 24576 + 16384) >> 15).astype(np.uint8))


def test_identities_and_half_turn():
    img = _img((41, 57, 3), 2)
    assert np.array_equal(O.rotate_image(0, img), img)
    assert np.array_equal(O.resize_image(1.0, img), img)
    assert np.array_equal(O.shear_image(0, img), img)
    assert np.array_equal(O.rotate_image(180, img), O.image_flip(-1, img))                     # odd sizes: (w // 2, h // 2) is the centre pixel
    assert np.array_equal(O.image_flip(0, img), img[::-1]) and np.array_equal(O.image_flip(1, img), img[:, ::-1])
    M = O.get_rotation_matrix_2d((28, 20), 30, 1.0)
    c, s = np.cos(np.pi / 6), np.sin(np.pi / 6)
    np.testing.assert_allclose(M, [[c, s, (1 - c) * 28 - s * 20], [-s, c, s * 28 + (1 - c) * 20]], rtol=0, atol=1e-12)


def test_resize_geometry_and_weights():
    img = _img((40, 56, 3), 3)
    for m in (0.9, 0.93, 1.07, 1.1):
        res = O.resize_cubic(img, m, m)
        assert res.shape == (int(np.rint(40 * m)), int(np.rint(56 * m)), 3)
        out = O.resize_image(m, img)
        assert out.shape == img.shape
        if m < 1:
            hs, ws = res.shape[:2]
            up, left = (40 - hs) // 2, (56 - ws) // 2
            assert np.array_equal(out[up:up + hs, left:left + ws], res)
            mask = np.ones((40, 56), bool)
            mask[up:up + hs, left:left + ws] = False
            assert not out[mask].any()
    # the cubic weights at fractions 1/4 and 3/4 (A = -0.75), 11 bits
    s, coef = O._cubic_axis(4, 0.5)                   # 2x zoom: fractions 0.75, 0.25, 0.75, 0.25
    assert s.tolist() == [-1, 0, 0, 1]
    assert coef[1].tolist() == [-216, 1800, 536, -72] and coef[0].tolist() == [-72, 536, 1800, -216]       # by hand: -0.10546875, 0.87890625, 0.26171875, -0.03515625
    flat = np.full((12, 16, 3), 200, np.uint8)
    assert np.array_equal(O.resize_cubic(flat, 1.5, 1.5), np.full((18, 24, 3), 200, np.uint8))


def test_float_and_int_vertical_passes_split_at_the_simd_width():
    """a row of the resized image has dw * C elements: the first 8 * (dw * C // 8) go through the float pass (VResizeCubicVec_32s8u),
    the tail through the int pass.  The two round the same sum and differ in fewer than one element per 1000 (float rounding next
    to a tie): tall narrow images make such elements appear on either side of the split."""
    for shape, nvec in (((30000, 7, 1), 0), ((30000, 8, 1), 8), ((20000, 5, 3), 8)):
        img = _img(shape, 5)
        f, i = O.resize_cubic_passes(img, 1.0, 1.07)
        assert np.abs(f - i).max() == 1 and (f != i).sum() < 1e-3 * f.size
        res = O.resize_cubic(img, 1.0, 1.07)
        dh = res.shape[0]
        flat = res.reshape(dh, -1)
        assert np.array_equal(flat[:, :nvec], f.reshape(dh, -1)[:, :nvec]) and np.array_equal(flat[:, nvec:], i.reshape(dh, -1)[:, nvec:])
        differ = (f != i).reshape(dh, -1)
        if nvec:
            assert differ[:, :nvec].any()
        if nvec < flat.shape[1]:
            assert differ[:, nvec:].any()
