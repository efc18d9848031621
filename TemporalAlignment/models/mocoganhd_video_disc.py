"""`TemporalAlignment.models.mocoganhd_video_disc.ModelD_3d` of the reference (:8-30) served by the gfx950 engine."""
from faceoff_amd.models.mocoganhd import ModelD_3d  # noqa: F401
