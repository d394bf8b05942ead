"""lstc_cls_concat_fwd[_pack] at the headline shape (N = 2048 sequences, S = 49, d = 2048): vector kernel vs the scalar kernel
(input 4 bytes off a 16-B boundary)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import _lib
from lstc_vad_amd._lib import dev_ptr, stream_ptr, check
lib = _lib.load()
DEV = "cuda"
N, S, d = 2048, 49, 2048
lo = torch.randn(N // 2, S - 1, d, device=DEV); hi = torch.randn(N // 2, S - 1, d, device=DEV)
off = torch.empty(lo.numel() + 1, device=DEV)[1:].view_as(lo); off.copy_(lo)
y = torch.empty(N, S, d, device=DEV)
buf = torch.empty(int(lib.lstc_pack1_bytes(N * S, d)), device=DEV, dtype=torch.uint8)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, x in (("vector", lo), ("scalar", off)):
    a = t(lambda: check(lib.lstc_cls_concat_fwd(dev_ptr(x), dev_ptr(hi), N // 2, None, None, dev_ptr(y), N, S, d, stream_ptr()), "f"))
    b = t(lambda: check(lib.lstc_cls_concat_fwd_pack(dev_ptr(x), dev_ptr(hi), N // 2, None, None, dev_ptr(y), N, S, d, dev_ptr(buf), stream_ptr()), "p"))
    gb = (N * (S - 1) * d * 4 + N * S * d * 4) / 1e9
    print(f"cls_concat {name}: f32 only {a:.0f} us ({gb / a * 1e3:.2f} TB/s), with pack {b:.0f} us ({(gb + N * S * d * 2 / 1e9) / b * 1e3:.2f} TB/s)")
