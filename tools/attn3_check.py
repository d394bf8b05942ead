"""Packed-input attention kernels (LstcAttnDesc.in_pack_cols > 0) against the f32-input bf16-mode kernels on the same
bf16-representable operands: probabilities, O and (backward) dQ | dK | dV; then timings of both forms.

    python tools/attn3_check.py [check|time] [S ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn, _lib
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
dev = "cuda"


def unpack1(buf, rows, K):
    """f32 [rows, K] of an lstc_pack1 buffer (csrc/lstc_common.h p1_offset)."""
    h = buf.view(torch.bfloat16)
    r = torch.arange(rows, device=buf.device).view(-1, 1)
    k = torch.arange(K, device=buf.device).view(1, -1)
    kbp = ((K + 63) // 64) * 2
    rr, kb, ch = r & 127, k >> 5, (k & 31) >> 3
    off = ((r >> 7) * kbp + kb) * 4096 + rr * 32 + ((ch ^ ((rr >> 2) & 3)) << 3) + (k & 7)
    return h[off.reshape(-1)].view(rows, K).float()


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    shapes = {49: (3, 256), 17: (1, 256), 81: (5, 256), 19: (2, 64), 33: (2, 64)}
    Ss = [int(a) for a in sys.argv[2:]] or [49, 17, 81, 19, 33]
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    for S in Ss:
        L, dk = shapes[S]
        dk = int(os.environ.get("A3_DK", dk))
        H = int(os.environ.get("A3_H", 8 if dk == 256 else 4))
        N = 2048 if mode == "time" else 256
        M = N * S
        g = torch.Generator(device=dev).manual_seed(5 + S)
        qkv = torch.randn(M, 3 * H * dk, device=dev, generator=g).bfloat16().float()
        do = torch.randn(M, H * dk, device=dev, generator=g).bfloat16().float()
        q, k, v = qkv[:, :H * dk], qkv[:, H * dk:2 * H * dk], qkv[:, 2 * H * dk:]
        idx = relative_position_index_3d(L, 4).to(dev) if S != 17 else None
        tab = torch.randn((2 * L - 1) * 49, H, device=dev, generator=g) * 0.3 if S != 17 else None
        qkv_p = Fn.pack3(qkv, False)
        do_p = Fn.pack3(do, False)
        pdrop = 0.2
        op_ref, pr_ref = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, pdrop, 7, packed=True)
        op_new, pr_new = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, tab, idx, pdrop, 7)
        torch.cuda.synchronize()
        o_ref, o_new = unpack1(op_ref.buf, M, H * dk), unpack1(op_new.buf, M, H * dk)
        dp = (pr_ref - pr_new).abs().max().item()
        do_ = (o_ref - o_new).abs().max().item()
        nbad = ((o_ref - o_new).abs() > 0.02 * o_ref.abs() + 1e-3).sum().item()
        print(f"FWD3 S={S} dk={dk}: probs max|d|={dp:.3e} (max {pr_ref.max().item():.3f}), O max|d|={do_:.3e} (max {o_ref.abs().max().item():.3f}), "
              f"O beyond 2%: {nbad} of {o_ref.numel()}, nan: {torch.isnan(o_new).sum().item()}", flush=True)
        if True:
            ref = Fn.attn_bwd(do, q, k, v, pr_ref, N, S, H, dk, dk, tab, idx, pdrop, 7, packed="fused")
            new = Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, tab, idx, pdrop, 7)
            torch.cuda.synchronize()
            g_ref, g_new = unpack1(ref[0].buf, M, 3 * H * dk), unpack1(new[0].buf, M, 3 * H * dk)
            for nm, c0 in (("dQ", 0), ("dK", H * dk), ("dV", 2 * H * dk)):
                a, b = g_ref[:, c0:c0 + H * dk], g_new[:, c0:c0 + H * dk]
                nb = ((a - b).abs() > 0.02 * a.abs() + 2e-3 * a.abs().max()).sum().item()
                print(f"BWD3 S={S} {nm}: max|d|={(a - b).abs().max().item():.3e} (max {a.abs().max().item():.3f}), beyond tol: {nb} of {a.numel()}, "
                      f"nan: {torch.isnan(b).sum().item()}", flush=True)
            if ref[3] is not None:
                print(f"BWD3 S={S} dtable: max|d|={(ref[3] - new[3]).abs().max().item():.3e} (max {ref[3].abs().max().item():.3f})", flush=True)
        if mode == "time":
            for rnd in range(2):
                t2, _ = timeit(lambda: Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, pdrop, 7, packed=True))
                t3, _ = timeit(lambda: Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, tab, idx, pdrop, 7))
                line = f"TIME S={S} N={N}: fwd f32-in {t2:.3f} ms, packed-in {t3:.3f} ms"
                if True:
                    b2, _ = timeit(lambda: Fn.attn_bwd(do, q, k, v, pr_ref, N, S, H, dk, dk, tab, idx, pdrop, 7, packed="fused"))
                    b3, _ = timeit(lambda: Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, tab, idx, pdrop, 7))
                    line += f"; bwd f32-in {b2:.3f} ms, packed-in {b3:.3f} ms"
                print(line, flush=True)
            for npw in (1, 2, 4, 8):
                Fn._ATTN_VARIANT = 100 + npw
                t3, _ = timeit(lambda: Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, tab, idx, pdrop, 7))
                print(f"TIME S={S} fwd packed-in, {npw} sequences per workgroup: {t3:.3f} ms", flush=True)
            Fn._ATTN_VARIANT = 0
            for npw in (2, 4, 8, 16):
                Fn._BWD_NPW = npw
                b3, _ = timeit(lambda: Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, tab, idx, pdrop, 7))
                print(f"TIME S={S} bwd packed-in, {npw} sequences per workgroup: {b3:.3f} ms", flush=True)
            Fn._BWD_NPW = 0


main()
