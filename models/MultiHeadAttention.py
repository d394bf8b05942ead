"""Reference import path ``models.MultiHeadAttention`` -> HIP-backed mirror (lstc_vad_amd.models.MultiHeadAttention)."""
from lstc_vad_amd.models.MultiHeadAttention import MultiHeadAttention  # noqa: F401
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_2d, relative_position_index_3d  # noqa: F401
