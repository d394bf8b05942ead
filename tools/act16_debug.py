"""Dev script: isolate the pieces of the bf16 activation stream (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_hip_parity as hp
from lstc_vad_amd import functional as Fn
DEV = torch.device("cuda", 0)
bf = lambda t: t.to(torch.bfloat16).float()
g = torch.Generator(device=DEV).manual_seed(5)
Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
for (M, N, K) in [(512, 256, 256), (2304, 2048, 512)]:
    x = torch.randn(M, K, device=DEV, generator=g); w = torch.randn(N, K, device=DEV, generator=g) * 0.1
    b = torch.randn(N, device=DEV, generator=g); r = bf(torch.randn(M, N, device=DEV, generator=g))
    rp = Fn.pack3(r, False)
    for kw in (dict(), dict(bias=b), dict(bias=b, dropout=(0.2, 0x1234567))):
        f32 = Fn.gemm(x, w, trans_b=True, residual=r, **kw)
        nores = Fn.gemm(x, w, trans_b=True, **kw)
        a = hp._unpack1(Fn.gemm(x, w, trans_b=True, residual=r, out_pack=True, **kw).buf, M, N)
        c = hp._unpack1(Fn.gemm(x, w, trans_b=True, residual=rp, out_pack=True, **kw).buf, M, N)
        torch.cuda.synchronize()
        print(M, N, K, list(kw), "opk+f32res vs f32:", float((a - bf(f32)).abs().max()), " opk+rpk vs f32:", float((c - bf(f32)).abs().max()),
              " rpk vs (nores+0):", float((c - bf(nores)).abs().max()), " rpk vs nores+2r", float((c - bf(nores + 2 * r)).abs().max()))
        bad = ((c - bf(f32)).abs() > 1e-2).nonzero()
        if len(bad):
            print("   first bad", bad[:6].tolist(), "n bad", len(bad), "of", M * N, " rows bad mod 16:", sorted(set((bad[:, 0] % 16).tolist())), "cols mod 64", sorted(set((bad[:, 1] % 64).tolist()))[:40])
Fn.set_x3_threshold()

# which residual values did the bad block receive?
Fn.set_x3_threshold(0, 0, 0)
M, N, K = 512, 256, 256
x = torch.randn(M, K, device=DEV, generator=g); w = torch.randn(N, K, device=DEV, generator=g) * 0.1
r = bf(torch.randn(M, N, device=DEV, generator=g)); rp = Fn.pack3(r, False)
zero = Fn.gemm(torch.zeros_like(x), w, trans_b=True, residual=rp, out_pack=True)     # product 0: the output IS the residual read
e = hp._unpack1(zero.buf, M, N)
bad = (e != r)
print("bad rows mod 128:", sorted(set((bad.nonzero()[:, 0] % 128).tolist())))
print("bad cols:", sorted(set((bad.nonzero()[:, 1]).tolist()))[:70])
for (i, j) in [(0, 0), (0, 4), (1, 0), (4, 0), (8, 0), (16, 0), (20, 8)]:
    hit = (r == e[i, j]).nonzero()
    print((i, j), "got", float(e[i, j]), "want", float(r[i, j]), "value found at", hit[:6].tolist())
hits = torch.zeros(1)
Fn.set_x3_threshold()
