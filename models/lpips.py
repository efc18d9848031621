"""`models.lpips.LPIPS` of the reference (models/lpips.py:53-93) on the gfx950 kernels: `LPIPS()(input, target)` ->
[N,1,1,1] per-image distances.  Weights are loaded with `load_state_dict` (reference key names); the reference's
download (lpips.py:12-48) has no counterpart here."""
from torch import nn

from faceoff_amd.loss import VQLPIPS as _VQLPIPS


class LPIPS(nn.Module):
    def __init__(self, use_dropout=True, dtype="fp32"):
        super().__init__()
        self._impl = _VQLPIPS(dtype=dtype)

    def load_state_dict(self, state_dict, strict=False):
        return self._impl.load_state_dict(state_dict, strict=strict)

    def state_dict(self, *a, **kw):
        return {k[len("perceptual_loss."):]: v for k, v in self._impl.state_dict(*a, **kw).items()}

    def forward(self, input, target):
        """[N,1,1,1], differentiable in `input` and `target` as the reference's (faceoff_amd.loss._LPIPSPerImageFunction)."""
        return self._impl.per_image(input, target)
