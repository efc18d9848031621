"""`scheduler.CycleScheduler` of the reference (scheduler.py:251-320; `from scheduler import CycleScheduler`,
train_faceoff_perceptual.py:14)."""
from faceoff_amd.scheduler import CycleScheduler  # noqa: F401

__all__ = ["CycleScheduler"]
