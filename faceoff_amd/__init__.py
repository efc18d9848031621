"""faceoff_amd: MI355X-native (gfx950) engine for the FaceOff VQ-VAE-2 + Conv3d-latent training step.

Host-side mirror of the reference interface for that path:
    faceoff_amd.models.vqvae_conv3d_latent.VQVAE / Quantize   (reference models/vqvae_conv3d_latent.py)
    faceoff_amd.distributed                                   (reference distributed/)
    faceoff_amd.scheduler.CycleScheduler                      (reference scheduler.py:251-320)
    faceoff_amd.trainer.FaceOffTrainer                        (reference train_faceoff_perceptual.py:32-47,92-107)
All arithmetic runs in faceoff_amd/libfaceoff_hip.so (C ABI: include/faceoff_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
