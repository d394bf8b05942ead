"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/lstc_hip.h
declares, host-side argument validation returns the documented error codes without touching a GPU, the model
mirror keeps the reference's constructor surface / state_dict keys, and the product path refuses CPU tensors
(there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from lstc_vad_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from lstc_vad_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "lstc_hip.h")).read()
    declared = set(re.findall(r"\b(lstc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    raw = C.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(raw, s), s
    assert lib.lstc_version() == 112        # 0.1.1 patch 2 (round 6): INTEGRATION.md, "ABI history"


def test_descriptor_layouts_match_header(lib, tmp_path):
    """The ctypes mirrors against the C compiler's own layout of include/lstc_hip.h: sizeof and the offset of every field of
    the three descriptors (a C program built with gcc prints them)."""
    import subprocess
    from lstc_vad_amd._lib import GemmDesc, AttnDesc, LossDesc, AdagradItem, VecItem, PackItem
    structs = {"LstcGemmDesc": GemmDesc, "LstcAttnDesc": AttnDesc, "LstcLossDesc": LossDesc, "LstcAdagradItem": AdagradItem,
               "LstcVecItem": VecItem, "LstcPackItem": PackItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lstc_hip.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), (cname, got[cname], C.sizeof(ct))
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)


def test_production_library_refuses_tuning_variants(lib):
    """LstcGemmDesc.variant is a public field: values outside the documented tile variants (and the timing-only ablation
    bits of -DLSTC_TUNING builds) must be refused before any launch, for every GEMM dtype."""
    from lstc_vad_amd._lib import GemmDesc
    d = GemmDesc()
    d.A = d.B = d.C = 16
    d.M = d.N = d.K = 128
    d.lda = d.ldb = d.ldc = 128
    d.transB = 1
    # f32: the production library holds 0 = 4 (default), 8 (its fallback), 11 (64x64 tail tile), 12 (persistent); the other tile variants
    # (1-3, 5-7, 9, 10) live in `make tuning` builds only since round 6
    for dtype, bad in ((0, (1, 2, 3, 5, 6, 7, 9, 10, 13, 14, 15, 16, 4 + 16, 1 << 20, -1)), (3, (1, 16, 32, (1 << 30) | 1))):      # LSTC_BF16P: 0 or LSTC_VARIANT_NO_QTAIL (1 << 30) alone
        d.dtype = dtype
        for v in bad:
            d.variant = v
            assert lib.lstc_gemm(C.byref(d), None) == -4, (dtype, v)
    assert lib.lstc_gemm_splits(0, 4224, 16) == 15 and lib.lstc_gemm_splits(2, 4224, 16) == 15       # 132 K tiles -> 15 slices of 9
    assert lib.lstc_gemm_splits(3, 100352, 8) == 8 and lib.lstc_gemm_splits(1, 130, 4) == 3 and lib.lstc_gemm_splits(0, 64, 1) == 1


def test_host_side_validation_error_codes(lib):
    from lstc_vad_amd._lib import GemmDesc, AttnDesc, LossDesc, VecItem
    assert lib.lstc_gemm(None, None) == -1
    # round-4 entry points: multi-tensor norm / clip and the batched column sums refuse bad arguments before any launch
    items = (VecItem * 2)(VecItem(16, 10000), VecItem(32, 5))
    assert lib.lstc_sqnorm_multi_scratch(items, 2) == 3 and lib.lstc_sqnorm_multi_scratch(None, 2) == 0      # ceil(10000 / 8192) + 1
    assert lib.lstc_sqnorm_multi(None, 2, 16, 3, 16, None) == -1 and lib.lstc_sqnorm_multi(items, 0, 16, 3, 16, None) == -2
    assert lib.lstc_sqnorm_multi(items, 2, 16, 2, 16, None) == -2                                             # scratch too small
    assert lib.lstc_clip_scale_multi(items, 2, None, 10.0, None) == -1 and lib.lstc_clip_scale_multi(items, 2, 16, 0.0, None) == -2
    assert lib.lstc_colsum_batched(None, 3, 8, 8, 8, 64, 16, 8, 16, None) == -1
    assert lib.lstc_colsum_batched(16, 0, 8, 8, 8, 64, 16, 8, 16, None) == -2 and lib.lstc_colsum_batched(16, 3, 8, 8, 4, 64, 16, 8, 16, None) == -2
    # round-5 entry points: the packed CLS passes, row unpacking and the K-chunk finish validate before any launch
    assert lib.lstc_cls_dot_pack(None, 16, 16, 16, 256, 49, 8, 2048, 1, 0.0, 0, None) == -1
    assert lib.lstc_cls_dot_pack(16, 16, 16, 16, 256, 49, 16, 2048, 1, 0.0, 0, None) == -4        # H > 8
    assert lib.lstc_cls_dot_pack(16, 16, 16, 16, 255, 49, 8, 2048, 1, 0.0, 0, None) == -4         # rows % 256
    assert lib.lstc_cls_dot_pack(16, 16, 16, None, 256, 49, 8, 2048, 1, 0.0, 0, None) == -1       # softmax mode needs probs
    assert lib.lstc_cls_wsum_pack(16, 16, 16, 256, 129, 8, 2048, None) == -5                      # S > 128
    assert lib.lstc_cls_outer_pack(16, 16, 16, 16, 8, 16, 256, 49, 8, 2048, None) == -3           # add0 not 16-B aligned
    assert lib.lstc_unpack1_rows(16, 256, 2048, 0, 49, 7, 16, 2048, None) == -2                   # rows past the end
    assert lib.lstc_unpack1_rows(16, 256, 2044, 0, 1, 4, 16, 2044, None) == -3                    # K % 8
    assert lib.lstc_splitk_finish(None, 4, 64, 8, 8, None, None, 0, None, 0, 16, 8, 0, 0.0, 0, 1, 0, 0, None) == -1
    assert lib.lstc_splitk_finish(16, 4, 32, 8, 8, None, None, 0, None, 0, 16, 8, 0, 0.0, 0, 1, 0, 0, None) == -2       # part_stride < M * N
    assert lib.lstc_splitk_finish(16, 4, 64, 8, 8, None, None, 0, None, 0, 16, 8, 1, 0.0, 0, 1, 0, 0, None) == -1       # BIAS without bias
    assert lib.lstc_splitk_finish(16, 4, 64, 8, 8, None, None, 0, None, 0, 16, 8, 128, 0.0, 0, 1, 0, 0, None) == -4     # pack flags
    assert lib.lstc_splitk_finish(16, 4, 64, 8, 8, 16, None, 0, None, 0, 16, 8, 1, 0.0, 0, 8, 256, 8, None) == -4       # groups: plain sums only
    assert lib.lstc_splitk_finish(16, 4, 64, 8, 6, None, None, 0, None, 0, 16, 8, 0, 0.0, 0, 1, 0, 0, None) == -3       # N % 4
    d = GemmDesc()
    assert lib.lstc_gemm(C.byref(d), None) == -1                      # NULL operands
    d.A = d.B = d.C = 16
    assert lib.lstc_gemm(C.byref(d), None) == -2                      # zero dims
    d.M = d.N = d.K = 8
    d.lda = d.ldb = d.ldc = 4
    assert lib.lstc_gemm(C.byref(d), None) == -2                      # ld < extent
    d.lda = d.ldb = d.ldc = 8
    d.dtype = 7
    assert lib.lstc_gemm(C.byref(d), None) == -4                      # unknown dtype
    a = AttnDesc()
    assert lib.lstc_attn_fwd(C.byref(a), None) == -1
    a.Q = a.K = a.V = a.O = a.probs = 16
    a.N, a.S, a.H, a.dk, a.dv = 1, 200, 1, 8, 8
    a.ldq = a.ldk = a.ldv = a.ldo = 8
    assert lib.lstc_attn_fwd(C.byref(a), None) == -5                  # S > 128
    # packed-input form (LstcAttnDesc.in_pack_cols, include/lstc_hip.h): every broken precondition is refused before any launch
    def packed_in():
        p = AttnDesc()
        p.Q = p.K = p.V = p.probs = p.O_pack = p.dO = p.dQ_pack = p.dK_pack = p.dV_pack = 4096
        p.N, p.S, p.H, p.dk, p.dv, p.dtype, p.scale = 256, 49, 4, 64, 64, 1, 0.125
        p.in_pack_cols, p.K_col0, p.V_col0, p.probs_ld = 768, 256, 512, 52
        p.dO_pack_cols, p.pack_cols, p.dK_col0, p.dV_col0 = 256, 768, 256, 512
        return p
    for field, value, fwd_rc, bwd_rc in (("dtype", 0, -4, -4), ("probs_ld", 49, -2, -2), ("probs_ld", 48, -2, -2), ("K_col0", 16, -2, -2),
                                         ("in_pack_cols", 800, -2, -2), ("V_col0", 640, -2, -2), ("O_pack", None, -1, None),
                                         ("dv", 32, -4, -4), ("S", 97, -4, -4), ("N", 255, -4, -4), ("Q", 4104, -3, -3), ("probs", 4100, -3, -3),
                                         ("dO_pack_cols", 0, None, -2), ("dO_pack_cols", 288, None, -2), ("dO", 4104, None, -3),
                                         ("dQ_pack", None, None, -1), ("dk", 96, None, -4)):
        p = packed_in()
        setattr(p, field, value)
        if fwd_rc is not None:
            assert lib.lstc_attn_fwd(C.byref(p), None) == fwd_rc, (field, value)
        if bwd_rc is not None:
            assert lib.lstc_attn_bwd(C.byref(p), None) == bwd_rc, (field, value)
    p = packed_in()
    p.in_pack_cols, p.probs_ld = 0, 52            # a padded probability pitch belongs to the packed-input kernels only
    p.O = 4096
    p.ldq = p.ldk = p.ldv = p.ldo = 256
    assert lib.lstc_attn_fwd(C.byref(p), None) == -4
    l = LossDesc()
    l.out = 16
    l.phase = 2
    assert lib.lstc_vad_loss(C.byref(l), None) == -1
    assert lib.lstc_layernorm_fwd(None, None, None, None, None, None, 1, 1, 1e-6, None) == -1
    assert lib.lstc_adagrad_step(16, 16, 16, 0, 0.1, 0.0, 1e-10, 1.0, None) == -2
    from lstc_vad_amd._lib import AdagradItem
    assert lib.lstc_adagrad_multi(None, 1, None) == -1
    it = (AdagradItem * 2)()
    it[0].w = it[0].grad = it[0].state = 16
    it[0].n = 8
    assert lib.lstc_adagrad_multi(it, 0, None) == -2
    assert lib.lstc_adagrad_multi(it, 2, None) == -1           # second item has NULL pointers
    it[1].w = it[1].grad = it[1].state = 16
    assert lib.lstc_adagrad_multi(it, 2, None) == -2           # ... and n = 0
    assert b"NULL" in lib.lstc_strerror(-1) and lib.lstc_strerror(0) == b"ok"


def test_model_mirror_keeps_reference_surface():
    from lstc_vad_amd.models import Encoder, Classifier, Regressor
    z = np.load(os.path.join(ROOT, "tests", "golden", "misc.npz"), allow_pickle=False)
    with torch.device("meta"):
        ltn = Encoder(3, 8, 256, 256, 2048, 4096, MHA_layerNorm=True, FFN_layerNorm=True, weight_init=False,
                      relative_pe=True, window_size=4, window_depth=3)
        stn = Encoder(3, 8, 256, 256, 2048, 3027, FFN_layerNorm=True, weight_init=False)
    assert list(ltn.state_dict().keys()) == z["ltn_state_keys"].tolist()
    assert list(stn.state_dict().keys()) == z["stn_state_keys"].tolist()
    assert sum(p.numel() for p in ltn.parameters()) == int(z["ltn_param_count"])
    assert sum(p.numel() for p in stn.parameters()) == int(z["stn_param_count"])
    assert list(Classifier(8).state_dict().keys()) == z["classifier_state_keys"].tolist()
    assert list(Regressor(8).state_dict().keys()) == z["regressor_state_keys"].tolist()
    # parameters the reference constructs but never uses get no gradient -> excluded from the all-reduce buckets
    assert len(stn.used_parameters()) == len(list(stn.parameters())) - 2 - 3 * 2
    assert len(ltn.used_parameters()) == len(list(ltn.parameters())) - 2


def test_relative_position_index_matches_reference_buffers():
    from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d, relative_position_index_2d
    z = np.load(os.path.join(ROOT, "tests", "golden", "misc.npz"), allow_pickle=False)
    for (L, ws) in [(3, 4), (2, 4), (5, 4), (2, 3), (1, 4)]:
        assert np.array_equal(relative_position_index_3d(L, ws).numpy(), z[f"relidx3d_L{L}_ws{ws}"])
    for ws in (3, 4):
        assert np.array_equal(relative_position_index_2d(ws).numpy(), z[f"relidx2d_ws{ws}"])


def test_no_cpu_fallback(lib):
    from lstc_vad_amd.models import Encoder
    enc = Encoder(1, 2, 4, 4, 8, 16)
    with pytest.raises(RuntimeError, match="HIP"):
        enc(torch.zeros(2, 4, 8))


def test_dropout_hash_matches_host_restatement():
    """The counter-based keep/drop rule is part of the ABI contract (masks are replayed by tests): restated in numpy
    (tests/util.py; the GPU suite compares lstc_dropout_mask with it bit for bit) - basic statistics here."""
    from util import host_dropout_keep as keep

    def key(p, seed):
        m = (1 << 64) - 1                               # splitmix64 finaliser over the 64-bit seed (csrc/lstc_common.h: drop_key_mix)
        zz = (seed + 0x9E3779B97F4A7C15) & m
        zz = ((zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9) & m
        zz = ((zz ^ (zz >> 27)) * 0x94D049BB133111EB) & m
        zz ^= zz >> 31
        return zz & 0xffffffff, zz >> 32, min(int(p * 4294967296.0), 0xffffffff)
    i = np.arange(1 << 20)
    for p in (0.1, 0.5, 0.6):
        m = keep(i, p, 0x1234567890ABCDEF)
        assert abs(m.mean() - (1 - p)) < 2e-3
        assert abs(np.corrcoef(m[:-1], m[1:])[0, 1]) < 5e-3
    # consecutive seeds (the sites of a step, the steps of a run: functional.next_seed is linear in its counter) give
    # independent masks - no two of them are shifted / XOR-permuted copies of one table: different (k0, k1) pairs
    # in BOTH words, and uncorrelated keep decisions at equal element indices
    base = 0x1234567890ABCDEF
    keys = [key(0.5, base + c)[:2] for c in range(64)]
    assert len({k[0] for k in keys}) == 64 and len({k[1] for k in keys}) == 64
    m0 = keep(i, 0.5, base)
    for c in (1, 2, 3, 17):
        assert abs(np.corrcoef(m0, keep(i, 0.5, base + c))[0, 1]) < 5e-3


def test_bf16p_epilogue_never_spills_a_register_with_a_load_in_flight():
    """tools/isa_guard.py: gemm_bf16p.hip issues epilogue operand loads through inline asm with hand-counted waits; the build is
    only correct if the register allocator leaves those destination registers alone until the wait (a spill there once made a
    landing load overwrite an address register: GPU memory fault).  Checked on the generated gfx950 assembly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_guard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("in-flight registers touched: []") >= 6, r.stdout
