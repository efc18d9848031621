"""`utils` of the reference for the hot path (utils.py:29-38,45-90): `process_data` and the model/loss factory the
trainer star-imports (`from utils import *`, train_faceoff_perceptual.py:17).  The dataset classes stay the reference's
(CPU image I/O, out of scope): the factory imports `TemporalAlignment.dataset` if it is importable and otherwise hands back
`None` loaders together with the model and loss, which is what the reference itself does when a loader cannot be built
(utils.py:68-75)."""
import os

import torch

from faceoff_amd.utils import process_data, split_batch  # noqa: F401

__all__ = ["process_data", "split_batch", "get_loaders_and_models", "get_facetranslation_latent_conv_perceptual",
           "save_frames_as_video", "save_image"]


def save_frames_as_video(frames, video_path, fps=30):
    """utils.py:9-17: `frames` = a list of [H,W,3] RGB float arrays in [0,1] (what `validation` builds,
    train_faceoff_perceptual.py:74-79), written as an mp4 through cv2 when cv2 is importable.  cv2 is not part of this
    image: the frames are then stored as `<video_path minus extension>.npy`, uint8 [T,H,W,3] (the same bytes the encoder
    would have been handed, `(frame*255).astype(np.uint8)`), so a validation pass never dies at its first clip."""
    import numpy as np
    u8 = [(np.asarray(f) * 255).astype(np.uint8) for f in frames]
    try:
        import cv2
    except ImportError:
        np.save(os.path.splitext(video_path)[0] + ".npy", np.stack(u8) if u8 else np.zeros((0, 0, 0, 3), np.uint8))
        return
    height, width, _ = u8[0].shape
    video = cv2.VideoWriter(video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (width, height))
    for frame in u8:
        video.write(cv2.cvtColor(frame, cv2.COLOR_RGB2BGR))
    video.release()


def save_image(data, saveas, video=False):
    """utils.py:19-26: `torchvision.utils.save_image(data, saveas, nrow=data.shape[0]//2, normalize=True, range=(-1, 1))`
    restated without torchvision: [N,C,H,W] in [-1,1] -> one image grid (2-pixel black padding, `nrow` images per row)."""
    from PIL import Image
    x = torch.as_tensor(data).detach().float().cpu()
    if x.dim() == 3:
        x = x.unsqueeze(0)
    if x.shape[1] == 1:
        x = x.expand(-1, 3, -1, -1)
    x = (x.clamp(-1.0, 1.0) + 1.0) / 2.0
    n, c, h, w = x.shape
    nrow = max(1, n // 2)
    xmaps, ymaps, pad = min(nrow, n), -(-n // min(nrow, n)), 2
    grid = torch.zeros((c, ymaps * (h + pad) + pad, xmaps * (w + pad) + pad))
    for k in range(n):
        r, q = divmod(k, xmaps)
        grid[:, r * (h + pad) + pad:r * (h + pad) + pad + h, q * (w + pad) + pad:q * (w + pad) + pad + w] = x[k]
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(saveas)


def _reference_loaders(args):
    """(train_loader, val_loader) over the reference's own TemporalAlignmentDataset when that package is importable (its
    cv2 / skimage / Wand dependencies are not part of this repository), else (None, None)."""
    try:
        from TemporalAlignment.dataset import TemporalAlignmentDataset
    except ImportError:
        return None, None
    from torch.utils.data import DataLoader
    opt = lambda name, default=None: getattr(args, name, default)
    common = dict(color_jitter_type=opt("colorjit"), grayscale_required=opt("gray", False))
    val_only = dict(cross_identity_required=opt("crossid", False), custom_validation_required=opt("custom_validation", False),
                    validation_datapoints=opt("validation_folder"))
    # one clip per batch (the frame axis is the model's batch axis, utils.py:69-80): up to 30 training / 50 validation frames
    loaders = []
    for mode, frames, extra, shuffle in (("train", 30, {}, True), ("val", 50, val_only, False)):
        loaders.append(DataLoader(TemporalAlignmentDataset(mode, frames, **common, **extra), batch_size=1, shuffle=shuffle, num_workers=2))
    return tuple(loaders)


def get_facetranslation_latent_conv_perceptual(args, device):
    """utils.py:45-83: the VQ-VAE (6 input channels) and the frozen LPIPS loss on `device`, plus the loaders."""
    from models.vqvae_conv3d_latent import VQVAE
    from loss import VQLPIPS
    model = VQVAE(in_channel=3 * 2).to(device)
    vqlpips = VQLPIPS(dtype=os.environ.get("FACEOFF_LPIPS_DTYPE", "fp32"))       # weights: loaded, never downloaded
    weights = os.environ.get("FACEOFF_LPIPS_WEIGHTS")
    if weights:
        vqlpips.load_state_dict(torch.load(weights, map_location="cpu"))
    train_loader, val_loader = _reference_loaders(args)
    return train_loader, val_loader, model, vqlpips.to(device)


def get_loaders_and_models(args, device):
    return get_facetranslation_latent_conv_perceptual(args, device)
