"""Validation / inference path of the reference trainer on the engine:
`validation(model, vqlpips, val_loader, device, epoch, i, sample_folder)` (train_faceoff_perceptual.py:53-79): every clip of
the validation loader through the model in eval mode without gradients, then source hulls, background, prediction and ground
truth de-normalised to 8-bit RGB frames (`(x.clamp(-1, 1) + 1) / 2`, :71-72; `(frame * 255).astype(np.uint8)`, utils.py:14)
and handed to a writer.  The reference's writer is cv2.VideoWriter (utils.py:9-17; cv2 is not available here): the default
writer stores `<name>.npy` arrays of shape [T,H,W,3]; pass `writer=` to plug an encoder in.

Also here: `MetricAccumulator`, the trainer's running `avg mse` (:108-121) without its per-step `.item()` host syncs and
pickled all_gather: two floats stay on the device and are summed over ranks with ONE all-reduce when somebody asks."""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from .utils import process_data


def denormalize_u8(frames, channels_last_ld=0, c0=0, bgr=False):
    """frames: NCHW [T,C,H,W] (channels c0..c0+2 are converted) or channels-last [T,H,W,ld] -> uint8 [T,H,W,3] on the device."""
    if channels_last_ld:
        T, H, W, ld = frames.shape
        Cc = 0
    else:
        frames = ops.dense_f32(frames, "frames")
        T, Cc, H, W = frames.shape
        ld = 0
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=frames.device)
    _lib.call("fo_denorm_u8", ops._ptr(frames), ld, Cc, c0, ops._ptr(out), T, H, W, int(bgr), ops._stream())
    return out


def npy_writer(frames_u8, path, fps=25):
    import numpy as np
    np.save(os.path.splitext(path)[0] + ".npy", frames_u8.cpu().numpy())


@torch.no_grad()
def validation(model, val_loader, device, epoch=0, global_step=0, sample_folder=None, writer=npy_writer, max_clips=None):
    """model: the drop-in VQVAE module (eval mode is set and restored here, :136,145).  Returns the list of per-clip dicts
    {name: uint8 [T,H,W,3]} for name in source, background, prediction, source_images, source_original (:62-68)."""
    was_training = model.training
    model.eval()
    results = []
    try:
        for i, data in enumerate(val_loader):
            if max_clips is not None and i >= max_clips:
                break
            img, S, ground_truth, source_original = process_data(data, device, None)
            out, _ = model(img)
            saves = {"source": denormalize_u8(img, c0=0), "background": denormalize_u8(img, c0=3), "prediction": denormalize_u8(out, c0=0),
                     "source_images": denormalize_u8(ground_truth), "source_original": denormalize_u8(source_original)}
            if sample_folder is not None and writer is not None:
                os.makedirs(sample_folder, exist_ok=True)
                for name, frames in saves.items():
                    writer(frames, f"{sample_folder}/{epoch + 1}_{global_step}_{i}_{name}.mp4", fps=25)
            results.append(saves)
    finally:
        model.train(was_training)
    return results


class MetricAccumulator:
    """mse_sum += recon_loss * S; mse_n += S every step (:111-121), on the device; `average()` = global mse_sum / mse_n."""

    def __init__(self, device):
        self.acc = torch.zeros(2, device=device)

    def update(self, recon_loss, S):
        self.acc[0:1] += recon_loss.detach().reshape(1) * float(S)
        self.acc[1:2] += float(S)

    def reduce_async(self, group=None):
        """Start the cross-rank sum (one 2-float all-reduce); returns (work handle or None, the tensor being reduced)."""
        t = self.acc.clone()
        from torch import distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(t, group=group, async_op=True), t
        return None, t

    def average(self, group=None):
        work, t = self.reduce_async(group)
        if work is not None:
            work.wait()
        s, n = t.tolist()
        return s / n if n else float("nan")
