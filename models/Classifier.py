"""Reference import path ``models.Classifier`` -> HIP-backed mirror (lstc_vad_amd.models.Classifier)."""
from lstc_vad_amd.models.Classifier import Classifier  # noqa: F401
