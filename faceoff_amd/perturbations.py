"""The cheap motion perturbations of the reference's data pipeline on the device: the affine
transformations of TemporalAlignment/perturbations.py (:45-105 translate_horizontal, translate_vertical, rotate_image,
resize_image; composition :208-264 perturb_image_composite, :271-295 perturb_image; ranges TemporalAlignment/ranges.py), applied
to whole clips [T,C,H,W] of float frames resident in HBM by one warp kernel (fo_affine_warp) instead of per frame with cv2 on
two CPU loader workers (utils.py:73).  Same function names, argument order and parameter draws (`random.randint` in the
reference's order, from a `random.Random`).

Differences: frames are float tensors on the GPU (the reference warps uint8 numpy images before normalisation);
interpolation is exact bilinear / bicubic -- cv2.warpAffine quantises source coordinates to 1/32 pixel, which is not
reproduced (cv2 is not available here: parity with cv2 is unpinned, the checker is oracle/warp_oracle.py); `distort_image`
(Wand arc / barrel distortions, :108-170) is not built."""
from __future__ import annotations

import ctypes as C
import math
import random as _random

import torch

from . import _lib, ops

translation_range = 3          # TemporalAlignment/ranges.py
rotation_range = 3
scale_ranges = (90, 110)


def _warp(frames, M, mode=0):
    """frames [T,C,H,W] (or [C,H,W]) float32 on the GPU; M = 2x3 FORWARD map (source pixel -> destination pixel), as the
    matrices handed to cv2.warpAffine: the kernel gets its inverse."""
    x = frames if frames.dim() == 4 else frames.unsqueeze(0)
    x = ops.dense_f32(x, "frames")
    a, b, tx, c, d, ty = (float(v) for v in M)
    det = a * d - b * c
    ia, ib, ic, id_ = d / det, -b / det, -c / det, a / det
    inv = (C.c_float * 6)(ia, ib, -(ia * tx + ib * ty), ic, id_, -(ic * tx + id_ * ty))
    out = torch.empty_like(x)
    T, Cc, H, W = x.shape
    _lib.call("fo_affine_warp", ops._ptr(x), ops._ptr(out), T, Cc, H, W, inv, mode, ops._stream())
    return out if frames.dim() == 4 else out[0]


def translate_horizontal(x, image):
    """perturbations.py:45-52: M = [[1, 0, x], [0, 1, 0]]."""
    return _warp(image, (1, 0, x, 0, 1, 0))


def translate_vertical(y, image):
    """:57-65: M = [[1, 0, 0], [0, 1, y]]."""
    return _warp(image, (1, 0, 0, 0, 1, y))


def rotation_matrix(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D: positive angle = counter-clockwise (origin top-left)."""
    al, be = scale * math.cos(math.radians(angle_deg)), scale * math.sin(math.radians(angle_deg))
    cx, cy = center
    return (al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy)


def rotate_image(rotation, image, center=None):
    """:70-82: about the image centre (w // 2, h // 2) or `center`."""
    h, w = image.shape[-2:]
    return _warp(image, rotation_matrix((w // 2, h // 2) if center is None else center, rotation))


def resize_image(magnification, image):
    """:87-105: cv2.resize(fx = fy = magnification, INTER_CUBIC) then centre crop (zoom in) or centre paste onto zeros (zoom
    out), as ONE bicubic warp: resized pixel u samples source (u + 0.5) / m - 0.5, and the crop / paste is an integer shift."""
    h, w = image.shape[-2:]
    m = float(magnification)
    ws, hs = int(round(w * m)), int(round(h * m))            # cv2.resize output size: saturate_cast<int>(size * f) = round
    if m >= 1:
        off_x, off_y = ws // 2 - w // 2, hs // 2 - h // 2    # destination x = resized x - off
        fwd = (ws / w, 0, (0.5 * ws / w - 0.5) - off_x, 0, hs / h, (0.5 * hs / h - 0.5) - off_y)
        return _warp(image, fwd, mode=1)
    off_x, off_y = (w - ws) // 2, (h - hs) // 2              # destination x = resized x + off; outside the pasted block: zeros
    fwd = (ws / w, 0, (0.5 * ws / w - 0.5) + off_x, 0, hs / h, (0.5 * hs / h - 0.5) + off_y)
    out = _warp(image, fwd, mode=1)
    mask = torch.zeros((h, w), device=out.device)
    mask[off_y:off_y + hs, off_x:off_x + ws] = 1.0
    return out * mask


def perturb_image(face_image, rng=None):
    """:271-295: ONE of {translate_horizontal, translate_vertical, rotate_image, resize_image} with the file's ranges."""
    rng = rng or _random
    fns = [translate_horizontal, translate_vertical, rotate_image, resize_image]
    ranges = {translate_horizontal: (-20, 20, 1), translate_vertical: (-20, 20, 1), rotate_image: (-25, 25, 1), resize_image: (90, 110, 100)}
    fn = fns[rng.randint(0, len(fns) - 1)]
    lo, hi, div = ranges[fn]
    return fn(rng.randint(lo, hi) / div, face_image)


def perturb_image_composite(face_image, eyes_center, rng=None):
    """:208-264 without `distort_image` (Wand): every perturbation is included with probability 1/2 (at least one), values drawn
    from TemporalAlignment/ranges.py; rotation is about the eye centre.  Returns (frames, gt_transformations)."""
    rng = rng or _random
    fns = [translate_horizontal, translate_vertical, rotate_image, resize_image]
    ranges = {translate_horizontal: (-translation_range, translation_range, 1), translate_vertical: (-translation_range, translation_range, 1),
              rotate_image: (-rotation_range, rotation_range, 1), resize_image: (scale_ranges[0], scale_ranges[1], 100)}
    gt = {"translate_horizontal": 0, "translate_vertical": 0, "rotate_image": 0}
    chosen = []
    while not chosen:
        chosen = [f for f in fns if rng.randint(0, 1)]
    for fn in chosen:
        lo, hi, div = ranges[fn]
        value = rng.randint(lo, hi) / div
        if fn is translate_horizontal:
            gt["translate_horizontal"] = value
        elif fn is translate_vertical:
            gt["translate_vertical"] = value
        else:
            gt["rotate_image"] = value               # (the reference records resize values under this key too, :253-254)
        face_image = fn(value, face_image, center=eyes_center) if fn is rotate_image else fn(value, face_image)
    return face_image, gt
