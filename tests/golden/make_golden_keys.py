#!/usr/bin/env python3
"""Key / shape / dtype lists of the reference's checkpoints (bare `state_dict` .pt files, train_faceoff_perceptual.py:143), taken
from the reference's own modules:  python tests/golden/make_golden_keys.py  ->  tests/golden/state_dict_keys.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("FACEOFF_REFERENCE", "/root/reference"))
from models.vqvae_conv3d_latent import VQVAE  # noqa: E402
from TemporalAlignment.models.mocoganhd_video_disc import ModelD_3d  # noqa: E402
from TemporalAlignment.models.mocoganhd_content_disc import ModelD_img  # noqa: E402

out = {}
for name, m in (("VQVAE(in_channel=6)", VQVAE(in_channel=6)),
                ("ModelD_3d(3,'instance',2,lr,False,16)", ModelD_3d(3, "instance", 2, 1e-4, False, 16)),
                ("ModelD_img(3,'instance',2,lr)", ModelD_img(3, "instance", 2, 1e-4))):
    out[name] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
json.dump(out, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0)
print({k: len(v) for k, v in out.items()})
