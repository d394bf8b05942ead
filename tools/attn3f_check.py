"""Exact-f32 attention forward on the lane-=-query structure (LstcAttnDesc.variant = 3 while under test) against the default
fp32 kernels: probabilities and O, then timings.   python tools/attn3f_check.py [S ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
dev = "cuda"


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


shapes = {49: 3, 17: 1, 81: 5, 19: 2, 33: 2, 64: 4, 96: 6}
for S in [int(a) for a in sys.argv[1:]] or [49, 17, 81, 33, 96]:
    L, H, dk, N = shapes[S], 8, 256, 2048
    M = N * S
    g = torch.Generator(device=dev).manual_seed(5 + S)
    q, k, v = (torch.randn(M, H * dk, device=dev, generator=g) for _ in range(3))
    idx = relative_position_index_3d(L, 4).to(dev) if S != 17 else None
    tab = torch.randn((2 * L - 1) * 49, H, device=dev, generator=g) * 0.3 if S != 17 else None
    Fn._ATTN_VARIANT = 0
    o0, p0 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7)
    Fn._ATTN_VARIANT = 3
    o3, p3 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7)
    torch.cuda.synchronize()
    print(f"F32-3 S={S}: probs max|d|={(p0 - p3).abs().max().item():.3e}, O max|d|={(o0 - o3).abs().max().item():.3e} (max {o0.abs().max().item():.2f}), nan {torch.isnan(o3).sum().item()}", flush=True)
    for rnd in range(2):
        Fn._ATTN_VARIANT = 0
        t0, _ = timeit(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7))
        Fn._ATTN_VARIANT = 3
        t3, _ = timeit(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7))
        print(f"TIME f32 S={S}: default {t0:.3f} ms, lane-=-query {t3:.3f} ms", flush=True)
    Fn._ATTN_VARIANT = 0
