"""Live pin of the CPU oracle against the imported reference (only where /root/reference exists: the build container;
the GPU box never sees the reference and skips this file).  Complements the committed golden vectors with fresh random
cases: different seeds, a different clip layout, eval mode."""
import os
import sys

import numpy as np
import pytest
import torch

from faceoff_amd.synth import make_state_dict, make_batch
from oracle import faceoff_oracle as O

REF = os.environ.get("FACEOFF_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "models", "vqvae_conv3d_latent.py")),
                                reason="the reference is not present on this machine")


def _ref_model(sd, train):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.vqvae_conv3d_latent import VQVAE                    # the reference's own module
    m = VQVAE(in_channel=6)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.train(train)
    return m


@pytest.mark.parametrize("train", [True, False])
def test_literal_forward_backward_equals_reference(train):
    """One clip of 3 frames at 32x48 through VQVAE.forward itself (T = N): outputs, indices, loss gradients, EMA buffers."""
    torch.manual_seed(0)
    sd = make_state_dict(21, codebook_scale=0.3, gain=2.0)
    img, gt = make_batch(22, 1, 3, 32, 48)
    x, y = torch.from_numpy(img), torch.from_numpy(gt).reshape(3, 3, 32, 48)
    m = _ref_model(sd, train)
    ids = {}
    m.quantize_t.register_forward_hook(lambda mod, a, o: ids.__setitem__("t", o[2]))
    m.quantize_b.register_forward_hook(lambda mod, a, o: ids.__setitem__("b", o[2]))
    dec, diff = m(x.reshape(3, 6, 32, 48))
    loss = torch.nn.functional.mse_loss(dec[:, :3], y) + diff.mean()
    loss.backward()
    p = O.to_torch_state(sd)
    r = O.run_step(x, torch.from_numpy(gt), p, training=train)
    r["loss"].backward()
    assert torch.equal(r["fw"]["id_t"], ids["t"]) and torch.equal(r["fw"]["id_b"], ids["b"])
    np.testing.assert_allclose(r["fw"]["dec"].detach().numpy(), dec.detach().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["loss"].item(), loss.item(), rtol=1e-6)
    for k, v in m.named_parameters():
        g = p[k].grad
        assert (g - v.grad).abs().max().item() <= 1e-5 * v.grad.abs().max().item() + 1e-12, k
    if train:
        for k, b in m.named_buffers():
            np.testing.assert_allclose(r["fw"]["new_buffers"][k].numpy(), b.numpy(), rtol=1e-5, atol=1e-7)
    else:
        assert r["fw"]["new_buffers"] is None
