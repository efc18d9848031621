"""Checkpoint compatibility with the reference: its checkpoints are bare `state_dict` .pt files
(train_faceoff_perceptual.py:143), loaded after stripping the DDP prefix `module.` (:178-185).  The mirrored modules must
expose exactly the reference's keys / shapes / dtypes in the reference's order (tests/golden/state_dict_keys.json, produced
from the reference's own modules) and survive a save / load round trip through a real file; where the reference is present
(build container) a checkpoint WRITTEN BY THE REFERENCE MODEL is loaded into the mirror and back."""
import json
import os
import sys

import pytest
import torch

REF = os.environ.get("FACEOFF_REFERENCE", "/root/reference")


def _mirrors():
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    from faceoff_amd.models.mocoganhd import ModelD_3d, ModelD_img
    return {"VQVAE(in_channel=6)": lambda: VQVAE(in_channel=6),
            "ModelD_3d(3,'instance',2,lr,False,16)": lambda: ModelD_3d(3, "instance", 2, 1e-4, False, 16),
            "ModelD_img(3,'instance',2,lr)": lambda: ModelD_img(3, "instance", 2, 1e-4)}


def test_state_dict_layout_equals_the_reference(golden_dir, tmp_path):
    want = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    for name, make in _mirrors().items():
        m = make()
        got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert got == want[name], name
        # a checkpoint saved from a DDP-wrapped model: every key prefixed with `module.` (:178-185 strips it)
        path = tmp_path / "ckpt.pt"
        torch.save({"module." + k: torch.randn(v.shape) if v.is_floating_point() else v for k, v in m.state_dict().items()}, path)
        sd = torch.load(path)
        sd = {k.replace("module.", ""): v for k, v in sd.items()}
        m2 = make()
        res = m2.load_state_dict(sd)
        assert not res.missing_keys and not res.unexpected_keys
        for k, v in m2.state_dict().items():
            assert torch.equal(v, sd[k]), k


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "models", "vqvae_conv3d_latent.py")), reason="the reference is not present")
def test_checkpoint_written_by_the_reference_model_round_trips(tmp_path):
    code = f"""
import sys, torch
sys.path.insert(0, {REF!r})
from models.vqvae_conv3d_latent import VQVAE
torch.manual_seed(3)
m = VQVAE(in_channel=6)
torch.save(m.state_dict(), {str(tmp_path / 'ref.pt')!r})
m2 = VQVAE(in_channel=6)
m2.load_state_dict(torch.load({str(tmp_path / 'mirror.pt')!r}), strict=True) if len(sys.argv) > 1 else None
print('REF_OK')
"""
    import subprocess
    # the reference's `models` package and this repository's shim share the name: run the reference in its own interpreter
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and "REF_OK" in out.stdout, out.stderr[-2000:]
    from faceoff_amd.models.vqvae_conv3d_latent import VQVAE
    sd = torch.load(tmp_path / "ref.pt")
    m = VQVAE(in_channel=6)
    m.load_state_dict(sd, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    torch.save(m.state_dict(), tmp_path / "mirror.pt")            # and back: the reference loads the mirror's checkpoint, strictly
    out = subprocess.run([sys.executable, "-c", code, "back"], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and "REF_OK" in out.stdout, out.stderr[-2000:]
