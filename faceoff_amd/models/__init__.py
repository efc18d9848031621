from .vqvae_conv3d_latent import VQVAE, Quantize  # noqa: F401
