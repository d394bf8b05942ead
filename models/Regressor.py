"""Reference import path ``models.Regressor`` -> HIP-backed mirror (lstc_vad_amd.models.Regressor)."""
from lstc_vad_amd.models.Regressor import Regressor  # noqa: F401
