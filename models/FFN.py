"""Reference import path ``models.FFN`` -> HIP-backed mirror (lstc_vad_amd.models.FFN)."""
from lstc_vad_amd.models.FFN import PositionwiseFeedForward  # noqa: F401
