"""CPU restatement of the OpenCV calls of the reference's perturbations -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/
may import it).

The reference's data pipeline (TemporalAlignment/perturbations.py) perturbs uint8 HxWx3 images with
    cv2.warpAffine(image, M, (w, h))                      :51 :63 :80 :117   (INTER_LINEAR, BORDER_CONSTANT 0)
    cv2.getRotationMatrix2D(center, rotation, 1.0)        :76 :78
    cv2.resize(image, None, fx=m, fy=m, INTER_CUBIC)      :88
    cv2.flip(image, flip_code)                            :125
The algorithm lives in a third-party dependency that is not part of /root/reference: OpenCV, pinned `opencv-python==4.6.0.66`
(environment.yml:70).  This file restates OpenCV 4.6.0's published algorithm for 8-bit images as numpy integer arithmetic:

  warpAffine (modules/imgproc/src/imgwarp.cpp: cv::warpAffine, hal::warpAffine, WarpAffineInvoker; remapBilinear with
  FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS>; initInterTab2D):
    * the 2x3 matrix is converted to double and inverted with the closed form D = 1 / (M0 M4 - M1 M3) ...
    * destination pixel (x, y) reads the source at fixed-point coordinates
          X = (saturate<int>((M1 y + M2) 1024) + 16 + saturate<int>(M0 x 1024)) >> 5        (1/32 pixel, AB_BITS 10, INTER_BITS 5)
      integer part X >> 5, fraction X & 31 (same for Y)
    * the four neighbours are blended with 15-bit integer weights (32 - fy)(32 - fx) 32 ..., the table entry (0, 0) being
      [32767, 0, 0, 1] (saturate_cast<short>(32768) and the sum fix-up), result (sum + 16384) >> 15
    * BORDER_CONSTANT 0: neighbours outside the image count as 0
  resize INTER_CUBIC, 8-bit (modules/imgproc/src/resize.cpp: hal::resize, HResizeCubic<uchar, int, short>,
  VResizeCubic<..., FixedPtCast<int, uchar, 22>, VResizeCubicVec_32s8u>):
    * dsize = (cvRound(w fx), cvRound(h fy)), scale = 1 / fx; destination column dx reads around sx = floor(fx_), fx_ =
      (float)((dx + 0.5) scale - 0.5); the four weights are interpolateCubic(fx_ - sx) (A = -0.75, float), times 2048 rounded
      to short; taps outside the row / column are clamped to the edge (replicate)
    * horizontal pass in int, vertical pass: the SIMD part of a row (the first 8 * (width * cn // 8) elements, 128-bit
      universal intrinsics in the SSE3-baseline wheel) in float -- S3 b3, then three mul+add, b = beta * 2^-22, round to nearest
      even -- the tail elements in int: (sum + 2^21) >> 22
  flip: an index reversal.

PARITY UNPINNED: cv2 is not installed here, so this restatement cannot be run against cv2 itself and the reference holds no
fixtures for these calls.  What it is pinned on: the properties the algorithm implies (tests/test_cv2_oracle_cpu.py: integer
translations are exact shifts, half-pixel translations are (a + b + 1) >> 1, rotation by 0 / zoom 1.0 are the identity, table sums).
Not restated: IPP's resize (used by the x86-64 wheel only when w * fx is a whole number, e.g. m = 1.0, where both are the identity);
non-x86 wheels may contract the float coefficient arithmetic."""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _round_int(v):
    """saturate_cast<int>(double) = cvRound: round half to even (cvtsd2si)"""
    return np.clip(np.rint(v), -2 ** 31, 2 ** 31 - 1).astype(np.int64)


def _sat_short(v):
    return np.clip(v, -32768, 32767)


def bilinear_tab():
    """BilinearTab_i [32*32][2][2] as initInterTab2D(INTER_LINEAR, fixpt) leaves it (imgwarp.cpp)."""
    one = np.float32(1.0)
    scale = np.float32(1.0 / INTER_TAB_SIZE)
    t1 = np.zeros((INTER_TAB_SIZE, 2), np.float32)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * scale
        t1[i] = (one - x, x)                                    # interpolateLinear
    tab = np.zeros(INTER_TAB_SIZE * INTER_TAB_SIZE * 4 + 8, np.int64)      # flat, as the static array (the fix-up loop reads past an entry)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            base = (i * INTER_TAB_SIZE + j) * 4
            isum = 0
            for k1 in range(2):
                for k2 in range(2):
                    v = t1[i, k1] * t1[j, k2]
                    q = int(_sat_short(np.rint(np.float32(v * np.float32(INTER_REMAP_COEF_SCALE)))))
                    tab[base + k1 * 2 + k2] = q
                    isum += q
            if isum != INTER_REMAP_COEF_SCALE:
                diff = isum - INTER_REMAP_COEF_SCALE
                ks2 = 1
                Mk = mk = (ks2, ks2)
                for k1 in range(ks2, ks2 + 2):
                    for k2 in range(ks2, ks2 + 2):
                        if tab[base + k1 * 2 + k2] < tab[base + mk[0] * 2 + mk[1]]:
                            mk = (k1, k2)
                        elif tab[base + k1 * 2 + k2] > tab[base + Mk[0] * 2 + Mk[1]]:
                            Mk = (k1, k2)
                if diff < 0:
                    tab[base + Mk[0] * 2 + Mk[1]] -= diff
                else:
                    tab[base + mk[0] * 2 + mk[1]] -= diff
    return tab[:INTER_TAB_SIZE * INTER_TAB_SIZE * 4].reshape(INTER_TAB_SIZE * INTER_TAB_SIZE, 4)


_BILINEAR = None


def get_rotation_matrix_2d(center, angle, scale):
    """cv::getRotationMatrix2D (imgwarp.cpp): center is a Point2f."""
    a = angle * (math.pi / 180)
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(M):
    """the inversion cv::warpAffine applies to a forward map (no WARP_INVERSE_MAP), in double"""
    M = [float(v) for v in np.asarray(M, np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def warp_affine(img, M):
    """cv2.warpAffine(img, M, (w, h)) for a uint8 image [H, W] or [H, W, C]; M = 2x3 forward map (any float dtype: converted to
    double like `M0.convertTo(matM, CV_64F)`)."""
    global _BILINEAR
    if _BILINEAR is None:
        _BILINEAR = bilinear_tab()
    img = np.asarray(img)
    assert img.dtype == np.uint8
    src = img if img.ndim == 3 else img[..., None]
    H, W, C = src.shape
    m = invert_affine(M)
    xs = np.arange(W, dtype=np.float64)
    ys = np.arange(H, dtype=np.float64)
    adelta = _round_int(m[0] * xs * AB_SCALE)
    bdelta = _round_int(m[3] * xs * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = _round_int((m[1] * ys + m[2]) * AB_SCALE) + round_delta
    Y0 = _round_int((m[4] * ys + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = _sat_short(X >> INTER_BITS)
    sy = _sat_short(Y >> INTER_BITS)
    w = _BILINEAR[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]      # [H, W, 4]

    def at(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)

    acc = (at(sy, sx) * w[..., 0:1] + at(sy, sx + 1) * w[..., 1:2] + at(sy + 1, sx) * w[..., 2:3] + at(sy + 1, sx + 1) * w[..., 3:4])
    out = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def _interpolate_cubic(x):
    """interpolateCubic (imgproc precomp.hpp), float arithmetic; x float32 array -> [..., 4] float32"""
    x = x.astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1)
    x1 = x + one
    c0 = ((A * x1 - np.float32(5) * A) * x1 + np.float32(8) * A) * x1 - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    xm = one - x
    c2 = ((A + np.float32(2)) * xm - (A + np.float32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def _cubic_axis(n_dst, scale):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    coef = _sat_short(np.rint(_interpolate_cubic(f) * np.float32(INTER_RESIZE_COEF_SCALE))).astype(np.int64)
    return s, coef


SIMD_LANES = 8          # v_uint16::nlanes of the 128-bit universal intrinsics


def resize_cubic_passes(img, fx, fy):
    """the resized image as the float vertical pass alone and as the int vertical pass alone would produce it ([dh, dw, C] each)"""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    src = img if img.ndim == 3 else img[..., None]
    H, W, C = src.shape
    dw, dh = int(_round_int(W * fx)), int(_round_int(H * fy))
    assert dw > 0 and dh > 0
    sx, alpha = _cubic_axis(dw, 1.0 / fx)
    sy, beta = _cubic_axis(dh, 1.0 / fy)
    s = src.astype(np.int64)
    hor = np.zeros((H, dw, C), np.int64)
    for j in range(4):
        hor += s[:, np.clip(sx - 1 + j, 0, W - 1), :] * alpha[None, :, j, None]
    rows = [hor[np.clip(sy - 1 + k, 0, H - 1)] for k in range(4)]                       # [dh, dw, C] each
    exact = sum(rows[k] * beta[:, k, None, None] for k in range(4))
    out_i = np.clip((exact + (1 << (2 * INTER_RESIZE_COEF_BITS - 1))) >> (2 * INTER_RESIZE_COEF_BITS), 0, 255)
    scale = np.float32(1.0 / (INTER_RESIZE_COEF_SCALE * INTER_RESIZE_COEF_SCALE))
    b = (beta.astype(np.float32) * scale).astype(np.float32)                             # [dh, 4]
    f = [r.astype(np.float32) for r in rows]
    acc = (f[3] * b[:, 3, None, None]).astype(np.float32)
    for k in (2, 1, 0):
        acc = ((f[k] * b[:, k, None, None]).astype(np.float32) + acc).astype(np.float32)
    out_f = np.clip(np.rint(acc), 0, 255).astype(np.int64)
    return out_f, out_i


def resize_cubic(img, fx, fy):
    """cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC) for a uint8 image."""
    img = np.asarray(img)
    out_f, out_i = resize_cubic_passes(img, fx, fy)
    dh, dw, C = out_f.shape
    width = dw * C
    nvec = width - width % SIMD_LANES
    flat_i = out_i.reshape(dh, width)
    flat_f = out_f.reshape(dh, width)
    out = np.concatenate([flat_f[:, :nvec], flat_i[:, nvec:]], 1).reshape(dh, dw, C).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def flip(img, flip_code):
    """cv2.flip: 0 = around the x axis (rows reversed), > 0 = around the y axis (columns reversed), < 0 both"""
    img = np.asarray(img)
    if flip_code == 0:
        return img[::-1].copy()
    if flip_code > 0:
        return img[:, ::-1].copy()
    return img[::-1, ::-1].copy()


# ---- the reference's perturbation functions over these calls (TemporalAlignment/perturbations.py)

def translate_horizontal(x, image):                      # :45-52
    return warp_affine(image, np.float32([[1, 0, x], [0, 1, 0]]))


def translate_vertical(y, image):                        # :57-65
    return warp_affine(image, np.float32([[1, 0, 0], [0, 1, y]]))


def rotate_image(rotation, image, center=None):          # :70-82
    h, w = image.shape[:2]
    c = (w // 2, h // 2) if center is None else center
    return warp_affine(image, get_rotation_matrix_2d(c, rotation, 1.0))


def resize_image(magnification, image):                  # :87-105
    res = resize_cubic(image, magnification, magnification)
    h, w = image.shape[:2]
    if magnification >= 1:
        cX, cY = res.shape[1] // 2, res.shape[0] // 2
        left, upper = cX - w // 2, cY - h // 2
        return res[upper:upper + h, left:left + w]
    out = np.zeros(image.shape, np.uint8)
    hs, ws = res.shape[:2]
    left, upper = (w - ws) // 2, (h - hs) // 2
    out[upper:upper + hs, left:left + ws] = res
    return out


def shear_image(shear, image):                           # :110-119
    return warp_affine(image, np.float32([[1, shear, 0], [shear, 1, 0]]))


def image_flip(flip_code, image):                        # :124-126
    return flip(image, int(flip_code))
