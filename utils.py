"""`utils` of the reference for the hot path (utils.py:29-38,45-90): `process_data` and the model/loss factory the
trainer star-imports (`from utils import *`, train_faceoff_perceptual.py:17).  The dataset classes stay the reference's
(CPU image I/O, out of scope): the factory imports `TemporalAlignment.dataset` if it is importable and otherwise hands back
`None` loaders together with the model and loss, which is what the reference itself does when a loader cannot be built
(utils.py:68-75)."""
import os

import torch

from faceoff_amd.utils import process_data, split_batch  # noqa: F401

__all__ = ["process_data", "split_batch", "get_loaders_and_models", "get_facetranslation_latent_conv_perceptual"]


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
