"""Mirror of the batch contract of the reference `utils.py:29-38` (`process_data`).

The loader yields `(source, target, background, source_images, source_images_original)`, each `[1,T,3,H,W]`
(batch_size=1 clips, TemporalAlignment/dataset.py).  `process_data` keeps the reference's signature and return value;
`split_batch` is the zero-copy form the trainer uses: it hands the model the (source, background) pair and lets the
input-layout kernel do the channel concatenation (`fo_nchw2_to_nhwc8`), so no `[T,6,H,W]` tensor is ever materialised.
Clip batches `[B,T,3,H,W]` (BASELINE configs 2-4) are accepted by both.
"""
from __future__ import annotations

import torch


def _frames(x):
    """[1,T,C,H,W] -> [T,C,H,W] (the reference's squeeze(0)); [B,T,C,H,W] -> [B*T,C,H,W]."""
    if x.dim() == 5:
        return x.reshape(-1, *x.shape[2:])
    return x


def process_data(data, device, dataset=None):
    """Reference semantics (utils.py:29-38): returns (img [T,6,H,W], S, ground_truth [T,3,H,W], source_images_original)."""
    source, target, background, source_images, source_images_original = data
    img = torch.cat([source, background], dim=2).squeeze(0).to(device)
    source = source.squeeze(0)
    ground_truth = source_images.squeeze(0).to(device)
    S = source.shape[0]
    return img, S, ground_truth, source_images_original.squeeze(0).to(device)


def split_batch(data, device):
    """-> ((source, background) each [N,3,H,W] on `device`, T, ground_truth [N,3,H,W]).  N = T, or B*T for clip batches.
    `source_images_original` is not moved: training never reads it."""
    source, target, background, source_images, _ = data
    T = source.shape[-4] if source.dim() >= 4 else source.shape[0]
    to = lambda x: _frames(x).to(device, dtype=torch.float32, non_blocking=True).contiguous()
    return (to(source), to(background)), T, to(source_images)
