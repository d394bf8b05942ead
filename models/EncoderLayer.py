"""Reference import path ``models.EncoderLayer`` -> HIP-backed mirror (lstc_vad_amd.models.EncoderLayer)."""
from lstc_vad_amd.models.EncoderLayer import EncoderLayer  # noqa: F401
