"""Reference import path ``models.Encoder`` -> HIP-backed mirror (lstc_vad_amd.models.Encoder)."""
from lstc_vad_amd.models.Encoder import Encoder  # noqa: F401
