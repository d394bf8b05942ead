"""MI355X-native LSTC_VAD training hot path: hand-written HIP kernels (liblstc_hip.so, C ABI in include/lstc_hip.h) behind the
reference's Python class surface (lstc_vad_amd.models) and Train/*.py command lines."""


def bump_weight_epoch():
    """Tell the GEMM layer that weights were rewritten behind autograd's back (``p.data.copy_(...)``, ``p.data.fill_``, an
    EMA through raw pointers ...): ``.data`` writes do not bump ``p._version``, so packed weight copies (f32x3 / bf16 GEMM
    modes) would otherwise be reused.  ``load_state_dict``, the optimizers and the weight-init helpers here call it
    themselves."""
    from .functional import bump_weight_epoch as _bump
    _bump()
