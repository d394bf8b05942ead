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
    # d_k = 32 at S = 81: the shape rules of the LDS-DMA staged attention kernels (d_k, d_v multiples of 32) hold, so this
    # case runs their 8-wave S > 64 instantiation inside a whole training step (every other reduced case has d_k = 16)
    "ltn_ubnormal_dk32": ("LTN", dict(d_model=32, n_head=2, d_k=32, d_v=32, d_inner=48, MHA_layerNorm=True,
                                      FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=5),
                          dict(batch_size=2, part_num=2, part_len=5, n_patch=16)),
}


# Full-width cases: the BASELINE model widths with enough tokens (T >= 4096) that the backward runs the
# production instantiations - the steady-state K loop of the exact-f32 GEMM in all three layouts, the batched split-K
# weight gradients, the d_k = 256 attention backward, the scalar-load path of the unaligned n_hidden = 3027.  Weights are
# 350-400 MB, so the fixtures hold the seed plus SAMPLES of the expected outputs (``sample_index``); both sides regenerate
# inputs and weights from lstc_vad_amd.synthetic (``fill_params``).
FULL_CASES = {
    "ltn_full": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                             FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3),
                 dict(batch_size=4, part_num=16, part_len=3, n_patch=16), 31),     # 128 sequences, S = 49, 6272 tokens
    "stn_full": ("STN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=3027, FFN_layerNorm=True),
                 dict(batch_size=2, part_num=16, part_len=4, n_patch=16), 32),     # 256 sequences, S = 17, 4352 tokens
    # BASELINE config 4 (UCF-Crime: n_patch = 9, part_num = 32, part_len = 2, README.md:59): S = 19, and the 3-D index is
    # built for a 4x4 window ([32, 32]) but only its top-left [18, 18] is read (models/MultiHeadAttention.py:107-111) -
    # the row stride of the index (32) differs from S - 1 (18) in the production attention instantiation (d_k = 256)
    "ltn_ucf_full": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                 FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=2),
                     dict(batch_size=4, part_num=32, part_len=2, n_patch=9), 33),   # 256 sequences, S = 19, 4864 tokens
    # BASELINE config 5's UBnormal half (README.md:55: --d_model 1024 --part_len 5): d_model = 1024 != H*d_k = 2048
    # (rectangular projections), S = 81 (the 8-wave staged attention instantiation at d_k = 256), table [441, 8]
    "ltn_ubnormal_full": ("LTN", dict(d_model=1024, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                      FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=5),
                          dict(batch_size=2, part_num=16, part_len=5, n_patch=16), 34),   # 64 sequences, S = 81, 5184 tokens
    # BASELINE config 3's third stage (Train/spatio_transformer_MIL_CE.py:23-44,156-181) at the width it is quoted on: the STN
    # (F = 3027, Regressor head) under the co-teaching loss - MIL with the flat-slice l1 quirk + the weighted BCE of the
    # part-mean scores against the pseudo labels (lambda_normal / lambda_abnormal, 1e-8 inside the logs)
    # --clip_grad (Train/temporal_transformer_shanghaitech.py:139-141) where it DOES something: at this width the encoder's
    # gradient norm is ~14 > 10, so clip_grad_norm_ really scales (the reduced-width clip case has norm 0.7: the coefficient clamps
    # to 1 there and only the norm computation is exercised)
    "ltn_clip_full": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                  FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3),
                      dict(batch_size=4, part_num=16, part_len=3, n_patch=16, clip_grad=True), 38),   # 128 sequences, S = 49
    "stn_mil_ce_full": ("STN_MIL_CE", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=3027, FFN_layerNorm=True),
                        dict(batch_size=2, part_num=16, part_len=4, n_patch=16), 37),   # 256 sequences, S = 17, 4352 tokens
}
# Full-width cases whose token count fills whole 256-row pack tiles (256 sequences): with the Q | K | V projections fused, the
# bf16 mode runs its attention core on PACKED operands (csrc/attention_pk.hip) at S = 49 (two query tiles) and S = 81 (three).
# Same recipe and fixture format as FULL_CASES; used by the GPU tests only (the CPU oracle suite keeps the smaller FULL_CASES).
PACKED_CASES = {
    "ltn_full_256": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                 FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3),
                     dict(batch_size=8, part_num=16, part_len=3, n_patch=16), 35),    # 256 sequences, S = 49, 12544 tokens
    "ltn_ubnormal_full_256": ("LTN", dict(d_model=1024, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                          FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=5),
                              dict(batch_size=8, part_num=16, part_len=5, n_patch=16), 36),   # 256 sequences, S = 81, 20736 tokens
}
# The step bench.py's ``value`` times, held by the reference itself (missing #2): B = 64 videos (--batch_size 32),
# T = 32 parts, P = 16 patches, d = 2048 - LTN at part_len 3 (2048 sequences of S = 49, 100 352 tokens,
# Train/temporal_transformer_shanghaitech.py:99-144) and the literal [64, 32, 16, 2048] STN input (2048 sequences of S = 17,
# Train/spatio_transformer_shanghaitech.py:90-101).  One reference step is 58 / 17 TFLOP and ~35 / 12 GB of autograd state on the
# CPU: ~7 + 2 minutes of the build container's 8 cores, so tests/test_golden_recipes.py regenerates these two only when
# LSTC_GOLDEN_HEADLINE=1 (the log of this round's regeneration: profiles/r06_headline_golden_regen.log); same recipe, same
# sampled fixture format as FULL_CASES.
HEADLINE_CASES = {
    "ltn_headline": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                 FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3),
                     dict(batch_size=32, part_num=32, part_len=3, n_patch=16), 41),   # 2048 sequences, S = 49, 100352 tokens
    "stn_headline": ("STN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=3027, FFN_layerNorm=True),
                     dict(batch_size=32, part_num=32, part_len=1, n_patch=16), 42),   # 2048 sequences, S = 17, 34816 tokens
    # BASELINE config 4 (the 8-GPU config: UCF-Crime, n_patch = 9, part_num = 32, part_len = 2) at the same global batch of 64 videos:
    # 2048 sequences of S = 19, the [32, 32] relative-position index read through [:18, :18]
    "ltn_ucf_headline": ("LTN", dict(d_model=2048, n_head=8, d_k=256, d_v=256, d_inner=4096, MHA_layerNorm=True,
                                     FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=2),
                         dict(batch_size=32, part_num=32, part_len=2, n_patch=9), 43),    # 2048 sequences, S = 19, 38912 tokens
}
N_SAMPLE = 256


def sample_index(numel, n=N_SAMPLE):
    """``n`` flat positions spread over a tensor by a multiplicative hash (portable, no RNG): the entries of a gradient
    / weight tensor that the full-width fixtures keep."""
    import numpy as np
    return (np.arange(min(n, numel), dtype=np.int64) * 2654435761 + 12345) % numel


def fill_params(module, seed):
    """Overwrite every parameter of a reference OR build module with portable-generator values (stream = position in
    ``named_parameters()``, identical on both sides)."""
    import torch
    from lstc_vad_amd import synthetic as syn
    with torch.no_grad():
        for i, (k, p) in enumerate(module.named_parameters()):
            if k.endswith("layer_norm.weight"):
                v = 1.0 + syn.small_uniform(p.shape, seed, 100 + i, 0.2)
            elif k.endswith("bias") and p.dim() == 1:
                v = syn.small_uniform(p.shape, seed, 100 + i, 0.1)
            elif k.endswith("relative_position_bias_table"):
                v = syn.small_uniform(p.shape, seed, 100 + i, 0.5)
            elif k in ("cls_token", "position_enc"):
                v = syn.small_uniform(p.shape, seed, 100 + i, 0.3)
            else:
                v = syn.xavier_uniform(p.shape, seed, 100 + i)
            p.copy_(torch.from_numpy(v))
