"""Hyper-parameters of the reference's `config.py` (:4-18) under the same names."""
DATASET = 11
LATENT_LOSS_WEIGHT = 1          # weight of the VQ commitment ("latent") loss, config.py:5
PERCEPTUAL_LOSS_WEIGHT = 1      # weight of the LPIPS term, config.py:6
# MoCoGAN-HD discriminator step (config.py:9-16)
G_LOSS_2D_WEIGHT = 0.25
G_LOSS_3D_WEIGHT = 0.25
image_disc_weight = 0.5
video_disc_weight = 0.5
D_LOSS_WEIGHT = 0.1
SAMPLE_SIZE_FOR_VISUALIZATION = 8
DISC_LOSS_WEIGHT = 0.25
