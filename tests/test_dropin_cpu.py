"""The reference trainer's unmodified import lines (train_faceoff_perceptual.py:14-18, utils.py:47-48,
models/vqvae_conv3d_latent.py:7) resolve to the MI355X engine when this repository's root is on sys.path."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_to_the_engine():
    code = r"""
import sys
sys.path.insert(0, %r)
from scheduler import CycleScheduler
import distributed as dist
import distributed as dist_fn
from utils import *
from config import DATASET, LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT, SAMPLE_SIZE_FOR_VISUALIZATION
from models.vqvae_conv3d_latent import VQVAE
from loss import VQLPIPS
import faceoff_amd.scheduler, faceoff_amd.loss, faceoff_amd.distributed, faceoff_amd.models.vqvae_conv3d_latent as mm
assert CycleScheduler is faceoff_amd.scheduler.CycleScheduler
assert VQVAE is mm.VQVAE and VQLPIPS is faceoff_amd.loss.VQLPIPS
for name in ("get_rank", "get_local_rank", "is_primary", "synchronize", "get_world_size", "all_reduce", "all_gather",
             "reduce_dict", "data_sampler", "LOCAL_PROCESS_GROUP", "launch"):
    assert hasattr(dist, name), name
assert dist.all_reduce is faceoff_amd.distributed.all_reduce and dist.launch is faceoff_amd.distributed.launch
assert (LATENT_LOSS_WEIGHT, PERCEPTUAL_LOSS_WEIGHT, SAMPLE_SIZE_FOR_VISUALIZATION, DATASET) == (1, 1, 8, 11)
assert callable(process_data) and callable(get_loaders_and_models)
m = VQVAE(in_channel=3 * 2)                       # utils.py:52 -- construction works anywhere; compute needs the GPU
assert len(m.state_dict()) == 76
print("IMPORTS_OK")
""" % ROOT
    # a fresh interpreter started OUTSIDE the repository, so only the sys.path line makes the names resolvable
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and "IMPORTS_OK" in out.stdout, out.stderr[-2000:]


def test_utils_shim_exports_every_name_the_reference_utils_defines(tmp_path):
    """`from utils import *` (train_faceoff_perceptual.py:17) must hand the trainers every function utils.py defines:
    validation() calls save_frames_as_video at its first clip (:79), the disc trainers call save_image."""
    names = ["save_frames_as_video", "save_image", "process_data", "get_facetranslation_latent_conv_perceptual", "get_loaders_and_models"]
    ref = "/root/reference/utils.py"
    if os.path.exists(ref):                       # live where the reference exists: its own list of top-level functions
        import ast
        names = [n.name for n in ast.parse(open(ref).read()).body if isinstance(n, ast.FunctionDef)]
        assert "save_frames_as_video" in names and "save_image" in names
    code = r"""
import sys
sys.path.insert(0, %r)
ns = {}
exec("from utils import *", ns)
missing = [n for n in %r if not callable(ns.get(n))]
assert not missing, missing
import numpy as np, torch
ns["save_frames_as_video"]([np.full((4, 6, 3), 0.5, np.float32)] * 3, %r, fps=25)
ns["save_image"](torch.zeros(4, 3, 5, 5), %r)
print("UTILS_OK")
""" % (ROOT, names, str(tmp_path / "clip.mp4"), str(tmp_path / "grid.png"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0 and "UTILS_OK" in out.stdout, out.stderr[-2000:]
    import numpy as np
    try:
        import cv2  # noqa: F401
        assert (tmp_path / "clip.mp4").exists()
    except ImportError:
        clip = np.load(tmp_path / "clip.npy")
        assert clip.shape == (3, 4, 6, 3) and clip.dtype == np.uint8 and int(clip[0, 0, 0, 0]) == 127
    assert (tmp_path / "grid.png").exists()
