"""CPU checker for the device-side perturbations -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ may import it).

numpy restatement of what `cv2.warpAffine(image, M, (w, h))` (INTER_LINEAR, BORDER_CONSTANT 0) and `cv2.resize(fx, fy,
INTER_CUBIC)` compute for the reference's perturbations (TemporalAlignment/perturbations.py:45-105), in exact arithmetic:
dst(x, y) = interpolate(src, M^-1 (x, y, 1)).  cv2 is not installed here (a dependency of the reference's data pipeline, pinned
`opencv-python` in environment.yml) so this cannot be checked against cv2 itself: PARITY UNPINNED.  Known deviation of the real
cv2: it quantises the source coordinates of a warp to 1/32 pixel and its fixed-point tables (imgwarp.cpp, INTER_BITS = 5)."""
import numpy as np


def _at(img, yy, xx):
    H, W = img.shape[-2:]
    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    return np.where(ok, img[..., np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)


def _cubic(t, k, A=-0.75):
    w_m1 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    w_0 = ((A + 2) * t - (A + 3)) * t * t + 1
    w_1 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    return {-1: w_m1, 0: w_0, 1: w_1, 2: 1.0 - w_m1 - w_0 - w_1}[k]


def warp_affine(img, M_forward, mode=0):
    """img [..., H, W] float; M_forward 2x3 (source -> destination), inverted here like cv2.warpAffine does."""
    H, W = img.shape[-2:]
    A = np.vstack([np.asarray(M_forward, np.float64).reshape(2, 3), [0, 0, 1]])
    inv = np.linalg.inv(A)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sx = (inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]).astype(np.float32)
    sy = (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]).astype(np.float32)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    tx, ty = sx - x0, sy - y0
    img = img.astype(np.float64)
    if mode == 0:
        return ((1 - ty) * ((1 - tx) * _at(img, y0, x0) + tx * _at(img, y0, x0 + 1)) +
                ty * ((1 - tx) * _at(img, y0 + 1, x0) + tx * _at(img, y0 + 1, x0 + 1)))
    out = 0.0
    for j in range(-1, 3):
        row = 0.0
        for i in range(-1, 3):
            row = row + _cubic(tx, i) * _at(img, y0 + j, x0 + i)
        out = out + _cubic(ty, j) * row
    return out
