"""`loss.VQLPIPS` of the reference (loss.py:27-33) served by the gfx950 LPIPS/VGG-16 kernels:
`from loss import VQLPIPS` in the reference trainer (utils.py:48) resolves here."""
from faceoff_amd.loss import VQLPIPS  # noqa: F401

__all__ = ["VQLPIPS"]
