"""Scan torch CPU thread counts for the oracle step (GPU-box host): picks a sane `cores` for cpu_baseline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
x = torch.randn(4096, 2048); w = torch.randn(4096, 2048)
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    (x @ w.t()).sum()
    t0 = time.perf_counter()
    for _ in range(5): y = x @ w.t()
    dt = (time.perf_counter() - t0) / 5
    print(f"threads {th}: sgemm 4096x2048x4096 {dt*1e3:.1f} ms = {2*4096*2048*4096/dt/1e12:.2f} TFLOP/s", flush=True)
