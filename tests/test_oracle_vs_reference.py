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


def test_composite_perturbation_draws_follow_the_references_random_stream():
    """faceoff_amd.perturbations.draw_composite against the reference's own perturb_image_composite
    (TemporalAlignment/perturbations.py:208-264, distort_image :131-165) run in a fresh interpreter with cv2 / wand / torchvision
    replaced by recorders: for 200 seeds the same perturbations with the same values, and the random stream left in the same state."""
    import json
    import random
    import subprocess
    from faceoff_amd import perturbations as P
    prog = r'''
import sys, types, json, random
sys.path.insert(0, %r)
class _Img:
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def distort(self, kind, args): CALLS.append(["distort_image", [TYPE[0], list(args[1:]) if kind != "arc" else list(args)], kind])
    def resize(self, *a): pass
    def __array__(self, *a, **k):
        import numpy as np
        return np.zeros((4, 4, 3), np.uint8)
for name in ("cv2", "wand", "wand.image", "torchvision", "torchvision.transforms", "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["wand.image"].Image = types.SimpleNamespace(from_array=lambda a: _Img())
sys.modules["PIL"].Image = sys.modules["PIL.Image"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
import numpy as np
import TemporalAlignment.perturbations as R
CALLS, TYPE = [], [None]
def rec(name):
    def f(v, image, **kw):
        CALLS.append([name, v]); return image
    return f
real_distort = R.distort_image
def distort(t, image):
    TYPE[0] = t
    return real_distort(t, image)
for n in ("translate_horizontal", "translate_vertical", "rotate_image", "resize_image"):
    setattr(R, n, rec(n))
R.distort_image = distort
R.find_eye_center = lambda lm: (1.0, 2.0)
out = []
for seed in range(200):
    random.seed(seed); del CALLS[:]
    _, gt = R.perturb_image_composite(np.zeros((4, 4, 3), np.uint8), None)
    out.append({"calls": [c[:2] for c in CALLS], "kinds": [c[2] for c in CALLS if len(c) > 2], "gt": gt, "next": random.random()})
print(json.dumps(out))
''' % REF
    res = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    ref = json.loads(res.stdout.strip().splitlines()[-1])
    n_distort = 0
    for seed, want in enumerate(ref):
        rng = random.Random(seed)
        plan = P.draw_composite(rng)
        got = [[n, [v[0], list(v[1])] if n == "distort_image" else v] for n, v in plan]
        # the reference hands ImageMagick (0.0, b, c, d): compare (type, b, c, d)
        want_calls = [[n, [v[0], v[1]]] if n == "distort_image" else [n, v] for n, v in want["calls"]]
        assert got == want_calls, (seed, got, want_calls)
        assert all(k == "barrel_inverse" for k in want["kinds"]), want["kinds"]      # the enum-tuple quirk: always the last branch
        assert rng.random() == want["next"], seed                                    # stream position after the call
        n_distort += any(n == "distort_image" for n, _ in plan)
    assert n_distort > 50
