"""`TemporalAlignment.models.mocoganhd_content_disc.ModelD_img` of the reference (:8-24) served by the gfx950 engine."""
from faceoff_amd.models.mocoganhd import ModelD_img  # noqa: F401
