"""`utils` of the reference for the hot path (utils.py:29-38,45-90): `process_data` and the model/loss factory the
trainer star-imports (`from utils import *`, train_faceoff_perceptual.py:17).  The dataset classes stay the reference's
(CPU image I/O, out of scope): the factory imports `TemporalAlignment.dataset` if it is importable and otherwise hands back
`None` loaders together with the model and loss, which is what the reference itself does when a loader cannot be built
(utils.py:68-75)."""
import os

import torch

from faceoff_amd.utils import process_data, split_batch  # noqa: F401

__all__ = ["process_data", "split_batch", "get_loaders_and_models", "get_facetranslation_latent_conv_perceptual"]


def get_facetranslation_latent_conv_perceptual(args, device):
    from torch.utils.data import DataLoader
    from models.vqvae_conv3d_latent import VQVAE
    from loss import VQLPIPS
    model = VQVAE(in_channel=3 * 2).to(device)                                   # utils.py:52
    vqlpips = VQLPIPS(dtype=os.environ.get("FACEOFF_LPIPS_DTYPE", "fp32"))       # utils.py:53; weights: no download here
    weights = os.environ.get("FACEOFF_LPIPS_WEIGHTS")
    if weights:
        vqlpips.load_state_dict(torch.load(weights, map_location="cpu"))
    vqlpips = vqlpips.to(device)
    train_loader = val_loader = None
    try:
        from TemporalAlignment.dataset import TemporalAlignmentDataset             # the reference's own dataset, if present
    except ImportError:
        return train_loader, val_loader, model, vqlpips
    train = TemporalAlignmentDataset("train", 30, color_jitter_type=getattr(args, "colorjit", None),
                                     grayscale_required=getattr(args, "gray", False))
    val = TemporalAlignmentDataset("val", 50, color_jitter_type=getattr(args, "colorjit", None),
                                   cross_identity_required=getattr(args, "crossid", False),
                                   grayscale_required=getattr(args, "gray", False),
                                   custom_validation_required=getattr(args, "custom_validation", False),
                                   validation_datapoints=getattr(args, "validation_folder", None))
    train_loader = DataLoader(train, batch_size=1, shuffle=True, num_workers=2)
    val_loader = DataLoader(val, batch_size=1, shuffle=False, num_workers=2)
    return train_loader, val_loader, model, vqlpips


def get_loaders_and_models(args, device):
    return get_facetranslation_latent_conv_perceptual(args, device)
