"""The cheap motion perturbations of the reference's data pipeline on the device: the affine
transformations of TemporalAlignment/perturbations.py (:45-105 translate_horizontal, translate_vertical, rotate_image,
resize_image; composition :208-264 perturb_image_composite, :271-295 perturb_image; ranges TemporalAlignment/ranges.py), applied
to whole clips [T,C,H,W] of float frames resident in HBM by one warp kernel (fo_affine_warp) instead of per frame with cv2 on
two CPU loader workers (utils.py:73).  Same function names, argument order and parameter draws (`random.randint` in the
reference's order, from a `random.Random`; `draw_composite` is checked live against the reference's own function).

Two frame formats, chosen by dtype:
  * uint8 [N,H,W,C] (or [H,W,C]) -- what cv2 hands the reference.  These go through the kernels of csrc/warp_u8.hip, which follow
    OpenCV 4.6.0's 8-bit arithmetic (the pinned opencv-python==4.6.0.66): 1/32-pixel source coordinates and 15-bit weights in
    warpAffine, 11-bit cubic weights and the int / float split of the vertical pass in resize, so the perturbed frame is the
    frame cv2 would have produced (checker: oracle/cv2_oracle.py; cv2 is not installed here, so parity with cv2 itself is
    unpinned).  Every function also takes one parameter PER FRAME (a sequence of N values): the reference draws them per image
    (TemporalAlignment/dataset.py:34-54).  `to_normalized` is the ToTensor + Normalize that follows (dataset.py:244-256).
  * float32 [T,C,H,W] -- already normalised clips: exact bilinear / bicubic interpolation (checker oracle/warp_oracle.py).
`distort_image` (Wand arc / barrel distortions, :131-168) is not built: ImageMagick is not restatable from the reference."""
from __future__ import annotations

import ctypes as C
import math
import random as _random
import struct

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


def _is_u8(image):
    return image.dtype == torch.uint8


def _f32(v):
    """the value after a round trip through float32 (np.float32([[1, 0, x], ...]), cv::Point2f)"""
    return struct.unpack("f", struct.pack("f", float(v)))[0]


def _u8_frames(image):
    x = image if image.dim() == 4 else image.unsqueeze(0)
    if not (x.is_cuda and x.dim() == 4 and x.shape[-1] <= 4):
        raise ValueError(f"faceoff_amd: uint8 frames must be [N,H,W,C] (or [H,W,C]) with C <= 4 on the GPU, got {tuple(image.shape)} on {image.device}")
    return x.contiguous()


def _per_frame(value, N):
    """one parameter for all N frames, or a sequence of N -> list of N"""
    if not hasattr(value, "__len__"):        # a number (Python or numpy scalar)
        return [float(value)], 0
    value = list(value)
    if len(value) != N:
        raise ValueError(f"faceoff_amd: {len(value)} per-frame parameters for {N} frames")
    return value, 1


def _warp_u8(image, matrices, per_frame):
    """cv2.warpAffine(frame, M, (w, h)) on every frame; matrices: list of 6-tuples (forward maps)"""
    x = _u8_frames(image)
    N, H, W, Cc = x.shape
    flat = [float(v) for M in matrices for v in M]
    out = torch.empty_like(x)
    _lib.call("fo_warp_affine_u8", ops._ptr(x), ops._ptr(out), N, H, W, Cc, (C.c_double * len(flat))(*flat), per_frame, ops._stream())
    return out if image.dim() == 4 else out[0]


def _hw(image):
    return tuple(image.shape[-3:-1]) if _is_u8(image) else tuple(image.shape[-2:])


def translate_horizontal(x, image):
    """perturbations.py:45-52: M = np.float32([[1, 0, x], [0, 1, 0]])."""
    if _is_u8(image):
        xs, pf = _per_frame(x, _u8_frames(image).shape[0])
        return _warp_u8(image, [(1, 0, _f32(v), 0, 1, 0) for v in xs], pf)
    return _warp(image, (1, 0, x, 0, 1, 0))


def translate_vertical(y, image):
    """:57-65: M = np.float32([[1, 0, 0], [0, 1, y]])."""
    if _is_u8(image):
        ys, pf = _per_frame(y, _u8_frames(image).shape[0])
        return _warp_u8(image, [(1, 0, 0, 0, 1, _f32(v)) for v in ys], pf)
    return _warp(image, (1, 0, 0, 0, 1, y))


def rotation_matrix(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D: positive angle = counter-clockwise (origin top-left)."""
    al, be = scale * math.cos(math.radians(angle_deg)), scale * math.sin(math.radians(angle_deg))
    cx, cy = center
    return (al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy)


def rotate_image(rotation, image, center=None):
    """:70-82: about the image centre (w // 2, h // 2) or `center` (uint8 frames: one centre, or one per frame)."""
    h, w = _hw(image)
    if _is_u8(image):
        N = _u8_frames(image).shape[0]
        rs, pf = _per_frame(rotation, N)
        one = center is not None and not hasattr(center[0], "__len__")       # (x, y) -- ints, floats or numpy scalars, as find_eye_center returns them
        cs = [(w // 2, h // 2)] if center is None else ([tuple(center)] if one else [tuple(c) for c in center])
        if len(cs) > 1 or pf:
            if len(cs) not in (1, N):
                raise ValueError(f"faceoff_amd: {len(cs)} rotation centres for {N} frames")
            rs, cs, pf = (rs * N if not pf else rs), (cs * N if len(cs) == 1 else cs), 1
        return _warp_u8(image, [rotation_matrix((_f32(c[0]), _f32(c[1])), r) for r, c in zip(rs, cs)], pf)     # center is a cv::Point2f
    return _warp(image, rotation_matrix((w // 2, h // 2) if center is None else center, rotation))


def resize_image(magnification, image):
    """:87-105: cv2.resize(fx = fy = magnification, INTER_CUBIC) then centre crop (zoom in) or centre paste onto zeros (zoom
    out), as ONE bicubic warp: resized pixel u samples source (u + 0.5) / m - 0.5, and the crop / paste is an integer shift.
    uint8 frames: OpenCV's fixed-point cubic resize + the crop / paste in one launch (fo_resize_center_u8)."""
    if _is_u8(image):
        x = _u8_frames(image)
        N, H, W, Cc = x.shape
        ms, pf = _per_frame(magnification, N)
        out = torch.empty_like(x)
        _lib.call("fo_resize_center_u8", ops._ptr(x), ops._ptr(out), N, H, W, Cc, (C.c_double * len(ms))(*[float(v) for v in ms]), pf, ops._stream())
        return out if image.dim() == 4 else out[0]
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


def shear_image(shear, image):
    """:110-119 (not in the reference's active lists): M = np.float32([[1, shear, 0], [shear, 1, 0]])."""
    if _is_u8(image):
        vs, pf = _per_frame(shear, _u8_frames(image).shape[0])
        return _warp_u8(image, [(1, _f32(v), 0, _f32(v), 1, 0) for v in vs], pf)
    return _warp(image, (1, shear, 0, shear, 1, 0))


def image_flip(flip_code, image):
    """:124-126 cv2.flip(image, int(flip_code)): 0 = rows reversed, > 0 = columns reversed, < 0 both."""
    code = int(flip_code)
    if _is_u8(image):
        x = _u8_frames(image)
        N, H, W, Cc = x.shape
        out = torch.empty_like(x)
        _lib.call("fo_flip_u8", ops._ptr(x), ops._ptr(out), N, H, W, Cc, code, ops._stream())
        return out if image.dim() == 4 else out[0]
    return torch.flip(image, [-2] if code == 0 else [-1] if code > 0 else [-2, -1])


def to_normalized(frames_u8, mean=0.5, std=0.5, reverse_channels=False):
    """transforms.ToTensor() + transforms.Normalize((mean,) * C, (std,) * C) (TemporalAlignment/dataset.py:244-256) on a stack of
    uint8 frames [N,H,W,C] -> float32 [N,C,H,W]; reverse_channels: BGR (cv2.imread) -> RGB."""
    x = _u8_frames(frames_u8)
    N, H, W, Cc = x.shape
    out = torch.empty((N, Cc, H, W), device=x.device, dtype=torch.float32)
    _lib.call("fo_u8_to_norm_nchw", ops._ptr(x), ops._ptr(out), N, H, W, Cc, int(bool(reverse_channels)), float(mean), float(std), ops._stream())
    return out if frames_u8.dim() == 4 else out[0]


def perturb_image(face_image, rng=None):
    """:271-295: ONE of {translate_horizontal, translate_vertical, rotate_image, resize_image} with the file's ranges."""
    rng = rng or _random
    fns = [translate_horizontal, translate_vertical, rotate_image, resize_image]
    ranges = {translate_horizontal: (-20, 20, 1), translate_vertical: (-20, 20, 1), rotate_image: (-25, 25, 1), resize_image: (90, 110, 100)}
    fn = fns[rng.randint(0, len(fns) - 1)]
    lo, hi, div = ranges[fn]
    return fn(rng.randint(lo, hi) / div, face_image)


N_DISTORTIONS = 3               # len(Distortion), TemporalAlignment/perturbations.py:36-39


def draw_composite(rng=None):
    """The random draws of `perturb_image_composite` (:236-262) in the reference's order, on the host: per pass FIVE coins
    (translate_horizontal, translate_vertical, rotate_image, resize_image, distort_image), repeated until one comes up; then per
    chosen perturbation its value.  A chosen `distort_image` draws its type from randint(0, len(Distortion)) and, inside the
    function, the three barrel_inverse parameters (:156-158: `Distortion.X.value` are 1-tuples -- `ARC = 1,` -- so the comparisons
    at :137,144 are never true and every type takes the last branch); those draws are burned here so that the stream stays the
    reference's.  Returns [(name, value), ...]; the distortion entry carries (type, (b, c, d))."""
    rng = rng or _random
    names = ("translate_horizontal", "translate_vertical", "rotate_image", "resize_image", "distort_image")
    ranges = {"translate_horizontal": (-translation_range, translation_range, 1), "translate_vertical": (-translation_range, translation_range, 1),
              "rotate_image": (-rotation_range, rotation_range, 1), "resize_image": (scale_ranges[0], scale_ranges[1], 100),
              "distort_image": (0, N_DISTORTIONS, 1)}
    chosen = []
    while not chosen:
        chosen = [n for n in names if rng.randint(0, 1)]
    plan = []
    for n in chosen:
        lo, hi, div = ranges[n]
        value = rng.randint(lo, hi) / div
        if n == "distort_image":
            value = (value, (rng.randint(0, 2) / 10, rng.randint(-5, 0) / 10, rng.randint(10, 10) / 10))
        plan.append((n, value))
    return plan


def perturb_image_composite(face_image, eyes_center, rng=None, distort_fn=None):
    """:208-264: every perturbation is included with probability 1/2 (at least one), values drawn from
    TemporalAlignment/ranges.py in the reference's order (`draw_composite`); rotation is about the eye centre.  `distort_image`
    (Wand / ImageMagick) is not built: its draws are consumed, and the frames pass through it unchanged unless the caller hands
    a `distort_fn(type, (b, c, d), frames)`.  Returns (frames, gt_transformations)."""
    fns = {"translate_horizontal": translate_horizontal, "translate_vertical": translate_vertical, "rotate_image": rotate_image,
           "resize_image": resize_image}
    gt = {"translate_horizontal": 0, "translate_vertical": 0, "rotate_image": 0}
    for name, value in draw_composite(rng):
        if name == "distort_image":
            gt["rotate_image"] = value[0]            # (the reference records resize and distortion values under this key too, :253-254)
            if distort_fn is not None:
                face_image = distort_fn(value[0], value[1], face_image)
            continue
        if name in ("translate_horizontal", "translate_vertical"):
            gt[name] = value
        else:
            gt["rotate_image"] = value
        fn = fns[name]
        face_image = fn(value, face_image, center=eyes_center) if name == "rotate_image" else fn(value, face_image)
    return face_image, gt
