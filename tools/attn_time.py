"""Times the fused attention kernels at the LTN-SHT layer shape (N=2048, S=49, H=8, dk=dv=256) with torch events."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
dev = "cuda"
for (N, S, L) in ((2048, 49, 3), (2048, 17, 1), (2048, 81, 5)):
    H, dk = 8, 256
    M = N * S
    q, k, v = (torch.randn(M, H * dk, device=dev) for _ in range(3))
    do = torch.randn(M, H * dk, device=dev)
    idx = relative_position_index_3d(L, 4).to(dev)
    tab = torch.randn((2 * L - 1) * 49, H, device=dev) * 0.1
    def t(fn, n=5):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): r = fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, r
    gb = 4 * M * H * dk * 4 / 1e9
    res = {}
    for rnd in range(2):                       # both kernel generations interleaved in ONE process (boxes differ by 2x)
        for variant in (0, 1):
            Fn._ATTN_VARIANT = variant
            tf, (o, p) = t(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7))
            tb, _ = t(lambda: Fn.attn_bwd(do, q, k, v, p, N, S, H, dk, dk, tab, idx, 0.2, 7))
            res.setdefault(variant, []).append((tf, tb))
    Fn.set_compute_dtype("bf16")               # bf16 mode: products on v_mfma_f32_32x32x16_bf16 (LstcAttnDesc.dtype = LSTC_BF16)
    for variant in (0, 1):
        Fn._ATTN_VARIANT = variant
        best = [1e9, 1e9]
        for rnd in range(2):
            tf, (o, p) = t(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7))
            tb, _ = t(lambda: Fn.attn_bwd(do, q, k, v, p, N, S, H, dk, dk, tab, idx, 0.2, 7))
            best = [min(best[0], tf), min(best[1], tb)]
        print(f"ATTN N={N} S={S} bf16 products gen {2 - variant}: fwd {best[0]:.3f} ms ({gb / best[0]:.2f} TB/s alg), bwd {best[1]:.3f} ms ({2 * gb / best[1]:.2f} TB/s alg)", flush=True)
    if S == 49:                                # bf16-mode forms: O / dQ|dK|dV written as packed bf16 operands (staged backward only)
        Fn._ATTN_VARIANT = 0
        for rnd in range(2):
            tf, (o, p) = t(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7, packed=True))
            tb, _ = t(lambda: Fn.attn_bwd(do, q, k, v, p, N, S, H, dk, dk, tab, idx, 0.2, 7, packed=True))
            tb3, _ = t(lambda: Fn.attn_bwd(do, q, k, v, p, N, S, H, dk, dk, tab, idx, 0.2, 7, packed="fused"))
            res.setdefault("pk", []).append((tf, tb, tb3))
        Fn.set_compute_dtype("fp32")
        print(f"ATTN N={N} S={S} packed outputs (bf16 products): fwd {min(x[0] for x in res['pk']):.3f} ms, bwd {min(x[1] for x in res['pk']):.3f} ms, "
              f"bwd fused pack {min(x[2] for x in res['pk']):.3f} ms", flush=True)
    Fn.set_compute_dtype("fp32")
    for variant in (0, 1):
        tf, tb = min(x[0] for x in res[variant]), min(x[1] for x in res[variant])
        print(f"ATTN N={N} S={S} gen {2 - variant}: fwd {tf:.3f} ms ({gb / tf:.2f} TB/s alg), bwd {tb:.3f} ms ({2 * gb / tb:.2f} TB/s alg)", flush=True)
