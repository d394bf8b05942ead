"""Reduced-width parity cases shared by make_golden.py (reference side) and the tests.

name -> (mode, reference ``Encoder`` keyword arguments, step arguments).  One case per
BASELINE.json config family plus the optional-flag and quirk cases of SURVEY.md 8c."""

CASES = {
    # name: (mode, encoder kwargs, step kwargs)
    "stn_sht": ("STN", dict(d_model=32, n_head=2, d_k=16, d_v=16, d_inner=47, FFN_layerNorm=True),
                dict(batch_size=2, part_num=3, part_len=2, n_patch=16)),
    "ltn_sht": ("LTN", dict(d_model=32, n_head=2, d_k=16, d_v=16, d_inner=64, MHA_layerNorm=True, FFN_layerNorm=True,
                            relative_pe=True, window_size=4, window_depth=3),
                dict(batch_size=2, part_num=3, part_len=3, n_patch=16)),
    "ltn_ucf": ("LTN", dict(d_model=32, n_head=2, d_k=16, d_v=16, d_inner=64, MHA_layerNorm=True, FFN_layerNorm=True,
                            relative_pe=True, window_size=4, window_depth=2),
                dict(batch_size=2, part_num=4, part_len=2, n_patch=9)),
    "ltn_ubnormal": ("LTN", dict(d_model=24, n_head=4, d_k=16, d_v=16, d_inner=40, MHA_layerNorm=True,
                                 FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=5),
                     dict(batch_size=1, part_num=3, part_len=5, n_patch=16)),
    "stn_mil_ce": ("STN_MIL_CE", dict(d_model=32, n_head=2, d_k=16, d_v=16, d_inner=47, FFN_layerNorm=True),
                   dict(batch_size=2, part_num=3, part_len=2, n_patch=16)),
    "stn_relpe2d_extras": ("STN", dict(d_model=32, n_head=2, d_k=16, d_v=8, d_inner=40, MHA_layerNorm=True,
                                       FFN_layerNorm=False, relative_pe_2D=True, window_size=4, CLS_learned=True,
                                       position_encoding=True, max_position_tokens=17, input_layerNorm=True),
                           dict(batch_size=2, part_num=2, part_len=2, n_patch=16)),
    "ltn_temporal_only_clip": ("LTN", dict(d_model=32, n_head=2, d_k=16, d_v=16, d_inner=64, MHA_layerNorm=False,
                                           FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3),
                               dict(batch_size=2, part_num=3, part_len=3, n_patch=16, temporal_only=True,
                                    clip_grad=True)),
}
