#!/usr/bin/env python3
"""Times the CLS-only layer's three passes over X: the f32 kernels (lstc_cls_dot / _wsum / _outer) against the packed-input ones
(lstc_cls_dot_pack / _wsum_pack / _outer_pack) at the headline shape (2048 x 49 x 2048, 8 heads) and UBnormal's (2048 x 81 x 1024)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn
dev = torch.device("cuda", 0)


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


Fn.set_compute_dtype("bf16")
for N, S, H, d in ((2048, 49, 8, 2048), (2048, 81, 8, 1024), (8960, 17, 8, 2048)):
    if (N * S) % 256:
        continue
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, S, d, device=dev, generator=g)
    u = torch.randn(N, H, d, device=dev, generator=g)
    w = torch.softmax(torch.randn(N, H, S, device=dev, generator=g), -1)
    xp = Fn.pack3(x.view(N * S, d), False)
    add0 = torch.randn(N, d, device=dev, generator=g)
    mb32, mb16 = N * S * d * 4 / 1e6, N * S * d * 2 / 1e6
    rows = [("dot (softmax + dropout)", lambda: Fn.cls_dot(u, x, 1, None, 0.2, 5), lambda: Fn.cls_dot_pack(u, xp, N, S, 1, None, 0.2, 5)),
            ("dot (backward)", lambda: Fn.cls_dot(u, x, 2, w, 0.2, 5), lambda: Fn.cls_dot_pack(u, xp, N, S, 2, w, 0.2, 5)),
            ("wsum", lambda: Fn.cls_wsum(w, x), lambda: Fn.cls_wsum_pack(w, xp, N, S)),
            ("outer", lambda: Fn.cls_outer(w, u, w, u, N, S, d), lambda: Fn.cls_outer_pack(w, u, w, u, add0, N, S, d))]
    print(f"N={N} S={S} H={H} d={d}: X {mb32:.0f} MB f32, {mb16:.0f} MB packed")
    for name, f32, pk in rows:
        a, b = t(f32), t(pk)
        print(f"  {name:26s} f32 {a:7.1f} us ({mb32 / a * 1e-0:6.0f} GB/s ... {mb32/a/1e3:.2f} TB/s)   pack {b:7.1f} us ({mb16/b/1e3:.2f} TB/s)")
Fn.set_compute_dtype("fp32")
