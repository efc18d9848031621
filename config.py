"""`config` of the reference (config.py:4-18): the constants the trainers import
(`from config import DATASET, LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT, SAMPLE_SIZE_FOR_VISUALIZATION`,
train_faceoff_perceptual.py:18; the GAN weights are read by disc_trainers/train_vqvae_mocoganhd_disc.py)."""
from faceoff_amd.config import *  # noqa: F401,F403
