"""Loss weights of the hot path (reference config.py:5-6)."""
LATENT_LOSS_WEIGHT = 1
PERCEPTUAL_LOSS_WEIGHT = 1
