"""Known-good reference point for the GEMM roofline discussion (cdna_hip_programming 5.4 rule 10): what the vendor
library (rocBLAS/hipBLASLt through torch.matmul) reaches for fp32 on the same shapes, same box, random data.
Measurement only; the product never calls it."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
def bench(M, N, K, ta, tb, iters=6):
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    A = a.t() if ta else a
    B = b.t() if tb else b
    for _ in range(6): c = A @ B
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = A @ B
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"VENDOR M={M} N={N} K={K} tA={ta} tB={tb}: {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
M = 100352
bench(M, 2048, 2048, 0, 1); bench(M, 4096, 2048, 0, 1); bench(M, 2048, 4096, 0, 1); bench(M, 2048, 2048, 0, 0)
bench(2048, 2048, M, 1, 0); bench(4096, 2048, M, 1, 0)
