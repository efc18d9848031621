"""`models.vqvae_conv3d_latent` of the reference (models/vqvae_conv3d_latent.py:33-83,192-295), served by the gfx950
engine: same class names, constructor arguments, methods and state_dict keys (faceoff_amd.models.vqvae_conv3d_latent)."""
from faceoff_amd.models.vqvae_conv3d_latent import VQVAE, Quantize  # noqa: F401

__all__ = ["VQVAE", "Quantize"]
